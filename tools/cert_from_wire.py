#!/usr/bin/env python
"""tools/cert_from_wire.py — SURVEY §8f rank 2 from the TRANSPORT's bytes (run on the GPU box): the ROUND-CHANGE messages of
one round change at N validators — Q = ⌊2N/3⌋+1 messages, each carrying a PreparedCertificate of 1 + (Q−1) messages:
Q·(Q+1) signatures — as protobuf bytes → the IsValidValidator verdict of every message, nested ones included.
  from_bytes : ONE ibft_verify_certificates_wire call (tree expansion, PayloadNoSig digests and signatures on the device)
  host_route : decode the messages, collect the nested ones, re-marshal PayloadNoSig of each and flatten on the host, then
               one ibft_verify_senders call (the route of f2 in round 1 and early round 2; C++ host code, one thread)
Prints one JSON object (p50 of `reps` calls each, ms; host columns → host-visible verdicts)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import go_ibft_amd.hostlib as H  # noqa: E402
import go_ibft_amd.verifier as V  # noqa: E402
import cert_cases as CC  # noqa: E402
from oracle import wire  # noqa: E402  (input generation only)
from oracle import workload as W  # noqa: E402

pinned = "--pinned" in sys.argv  # the wire bytes in an ibft_pinned_alloc buffer (what a transport that receives into such buffers hands over)
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [64, 256]
out = {"wire_memory": "ibft_pinned_alloc" if pinned else "ordinary (pageable)", "note": "wire bytes in ordinary host memory unless wire_memory says otherwise; every form ends with host-visible verdicts; keys: cold = no key known, "
               "warm = IBFT_FLAG_PUBKEY_CACHE after one pass"}
for n in sizes:
    reps = 30 if n <= 256 else 6
    r = W.make_round(n, 900 + n, height=5, round_=1, raw_len=1024)
    q = (2 * n) // 3 + 1
    pm = CC.preprepare(r, 1, 5, 1)
    prepares = [CC.prepare(r, j, 5, 1) for j in range(n) if j != 1][: q - 1]
    pcb = wire.prepared_certificate(pm, prepares)
    rcs = [CC.round_change(r, i, 5, 2, wire.Proposal(r.raw, 1), pcb).encode() for i in range(q)]
    buf, off = CC.pack(rcs)
    if pinned:
        buf = V.pinned_copy(np.frombuffer(buf, dtype=np.uint8))
    rows_expected = q * (q + 1)
    res = {"validators": n, "round_change_messages": q, "signatures": rows_expected, "wire_bytes": len(buf)}
    for name, flags in (("cold", 0), ("warm", V.FLAG_PUBKEY_CACHE)):
        bv = V.BatchVerifier(flags=flags, max_rows=max(65536, rows_expected + 64))
        bv.set_validators(5, r.addrs, r.power)
        form = {}
        for label, route in (("from_bytes", 0), ("host_route", 1)):
            for _ in range(3):
                rows, valid, _, _ = H.cert_routes(bv, buf, off, route, rows_expected + 64)
            assert rows == rows_expected and valid == rows_expected, (label, rows, valid)
            t, hm = [], []
            for _ in range(reps):
                _, _, host_ms, total_ms = H.cert_routes(bv, buf, off, route, rows_expected + 64)
                t.append(total_ms)
                hm.append(host_ms)
            p = np.percentile(t, [10, 50, 90])
            form[label] = {"p50_ms": round(float(p[1]), 3), "p10_ms": round(float(p[0]), 3), "p90_ms": round(float(p[2]), 3),
                           "signatures_per_s": round(rows_expected / (p[1] * 1e-3)),
                           **({"host_prepare_ms_p50": round(float(np.percentile(hm, 50)), 3)} if route else {})}
        res[name] = form
        bv.close()
    out[f"n{n}"] = res
print(json.dumps(out, indent=1))
