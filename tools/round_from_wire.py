#!/usr/bin/env python
"""tools/round_from_wire.py — BASELINE config #3 from the TRANSPORT's bytes (run on the GPU box): the 4 095 PREPARE and
4 096 COMMIT messages of one height as protobuf bytes → sender verdicts, closure verdicts (a1, a1 ∧ a2) and the COMMIT
quorum, with no proto.Unmarshal / PayloadNoSig / flattening on the host.
  two_step : per set ibft_verify_senders_wire (+ ibft_wire_stage_seals + ibft_seals_run for the COMMIT set) and
             ibft_verify_hashes over the hash column the walk returned — the route of rounds 1–2
  sets     : one ibft_verify_messages_wire call per set
Prints one JSON object (p50 of 200 rounds, ms)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import go_ibft_amd.verifier as V  # noqa: E402
import wire_cases as WCASE  # noqa: E402
from oracle import workload as W  # noqa: E402  (input generation only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = 200
r = W.make_round(n, 800 + n, height=5, round_=0)
commits = WCASE.canonical_round(r, ("commit",))
prepares = WCASE.canonical_round(r, ("prepare",))[1:]
cw, coff = WCASE.pack(commits)
pw, poff = WCASE.pack(prepares)
cw, coff, pw, poff = (V.pinned_copy(x) for x in (cw, coff, pw, poff))
out = {"rows": n, "wire_bytes": {"prepare": int(len(pw)), "commit": int(len(cw))}, "rounds": rounds,
       "note": "wire bytes in ibft_pinned_alloc buffers; host-visible verdict masks + quorum flag at the end of each form; the 1 KiB proposal is hashed once per round"}
for name, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    bv.set_validators(r.height, r.addrs, r.power)
    l32p, l32c = np.full(n - 1, 32, np.uint8), np.full(n, 32, np.uint8)

    def two_step():
        bv.forget_proposal()            # a new height has a new proposal: hashed once per round
        s1, rows, _ = bv.is_valid_validator_wire(pw, poff)
        h1 = bv.is_valid_proposal_hash(r.raw, r.round, rows["proposal_hash"], l32p)
        s2, rows, _ = bv.is_valid_validator_wire(cw, coff)
        bv.wire_stage_seals()
        a2, t = bv.seals_run()
        h2 = bv.is_valid_proposal_hash(r.raw, r.round, rows["proposal_hash"], l32c)
        return s1.all() and h1.all() and s2.all() and a2.all() and h2.all(), t

    def sets():
        bv.forget_proposal()
        s1, v1, _, _ = bv.verify_messages_wire(pw, poff, r.height, r.round, raw=r.raw, want_rows=False)
        s2, v2, _, t = bv.verify_messages_wire(cw, coff, r.height, r.round, raw=r.raw, want_rows=False)
        return s1.all() and v1.all() and s2.all() and v2.all(), t
    res = {}
    for form, fn in (("two_step", two_step), ("sets", sets)):
        for _ in range(3):
            ok, t = fn()
        assert ok and t.has_quorum == 1 and t.valid_rows == n, (form, ok, t.valid_rows)
        lat = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            fn()
            lat.append(time.perf_counter() - t0)
        q = np.percentile(np.array(lat) * 1e3, [10, 50, 90])
        res[form] = {"p50_ms": round(float(q[1]), 4), "p10_ms": round(float(q[0]), 4), "p90_ms": round(float(q[2]), 4)}
    out[name] = res
    bv.close()
print(json.dumps(out, indent=1))
