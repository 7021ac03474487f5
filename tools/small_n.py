#!/usr/bin/env python3
"""tools/small_n.py — what a node with the reference's OWN validator counts gets (round-5 review, item 6): one round of
COMMIT seals at N = 4, 6, 30 (core/consensus_test.go:139, core/byzantine_test.go:21, core/rapid_test.go:156) and 8 … 256,
host columns → host-visible verdicts through ibft_verify_seals (H2D, launch, kernel, tally, D2H: what a BatchVerifier call
costs), cold (every row recovered) and warm (keys known), p50 of 300 calls behind 100 untimed ones — next to the TUNED CPU
recovery (oracle/recover_tuned.inc, libsecp256k1-class) on ONE core for the same rows, which is what the Go closure of
core/ibft.go:932-944 costs with go-ethereum's secp256k1 behind IsValidCommittedSeal.  The crossover is the smallest N at which
the device call is faster than that loop: the value for IBFT_MIN_DEVICE_ROWS (shim/go/core/backend_batch.go, backend.hpp).
Prints one JSON object.  The CPU leg uses the oracle (test infrastructure) as the CPU baseline, like bench.py's cpu_baseline."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd.numa as NUMA  # noqa: E402
pin = NUMA.pin_to_device_node(0)
import go_ibft_amd.verifier as V  # noqa: E402
from oracle import binding as B, workload as W  # noqa: E402

SIZES = [int(x) for x in sys.argv[1:]] or [4, 6, 8, 12, 16, 30, 64, 128, 256]
out = {"definition": "p50 ms of ONE ibft_verify_seals call (pinned host columns -> host-visible verdict words + tally) on N rows, "
                     "cold = key cache off, warm = keys known; cpu_one_core_ms = the same rows through the tuned CPU recovery on one "
                     "thread (p50 of 30); crossover = smallest N with device < cpu",
       "numa_pin": pin, "sizes": []}
ctx = {"cold": V.BatchVerifier(flags=0, max_rows=1024), "warm": V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE, max_rows=1024)}
try:
    out["device_canary_ns"] = ctx["cold"].issue_probe()[0]
    for n in SIZES:
        r = W.make_round(n, 4000 + n)
        vs = B.ValSet(r.addrs, r.power)
        cols = tuple(V.pinned_copy(x) for x in (r.hash32, r.seal65, r.signer20))
        ent = {"validators": n}
        for name, bv in ctx.items():
            bv.set_validators(n, r.addrs, r.power)
            for _ in range(100):
                got, t = bv.is_valid_committed_seal(*cols)
            assert got.all() and t.has_quorum == 1 and t.distinct_senders == n
            lat = []
            for _ in range(300):
                t0 = time.perf_counter()
                bv.is_valid_committed_seal(*cols)
                lat.append(time.perf_counter() - t0)
            ent[name + "_ms_p50"] = float(np.median(lat) * 1e3)
            ent[name + "_dispatch_cold_warm_lanes"] = [int(x) for x in bv.last_dispatch()]
        cpu = []
        for _ in range(30):
            t0 = time.perf_counter()
            v = B.verify_seals_tuned(vs, r.hash32, r.seal65, r.signer20, nthreads=1)
            cpu.append(time.perf_counter() - t0)
        assert v.all()
        ent["cpu_one_core_ms"] = float(np.median(cpu) * 1e3)
        ent["cpu_us_per_verify"] = ent["cpu_one_core_ms"] * 1e3 / n
        out["sizes"].append(ent)
finally:
    for bv in ctx.values():
        bv.close()
for name in ("cold", "warm"):
    cross = next((e["validators"] for e in out["sizes"] if e[name + "_ms_p50"] < e["cpu_one_core_ms"]), None)
    out["crossover_" + name] = cross
print(json.dumps(out))
