#!/usr/bin/env python3
"""tools/table_ab.py — where the lane / group cold kernels keep their window tables, same box, one process (round-4 review, item
4): IBFT_COLD_TABLE = lds (round 5: the workgroup's LDS, no private segment) / private (round 4: private segment, entries read in
front of the doublings) / private2 (private segment, no prefetch: two resident wavefronts per SIMD; lane kernel only), at the
sizes these kernels serve.  Contexts alternate; kernel ms by HIP events behind ≥ 120 untimed passes.

    python tools/table_ab.py > gpurun_out/profiles/r05_table_ab.txt"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd.numa as NUMA  # noqa: E402
print("#", NUMA.pin_to_device_node(0))
import go_ibft_amd.verifier as V  # noqa: E402
import go_ibft_amd.simulate as SIM  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768, 65536, 131072, 262144]
print("# rows  lanes  table     kernel ms (median of rounds)   ns/verify   M verifies/s (step)")
for n in sizes:
    ctx = {}
    for tab in ("lds", "private", "private2"):
        os.environ["IBFT_COLD_TABLE"] = tab
        bv = V.BatchVerifier(flags=0, max_rows=n)
        if not ctx:
            r = SIM.make_round(bv, n, 900 + n)
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
        verdict, t = bv.seals_run()
        assert verdict.all() and t.has_quorum == 1
        lanes = bv.last_dispatch()[0]
        if tab == "private2" and lanes != 1:
            bv.close()
            continue
        ctx[tab] = bv
    os.environ.pop("IBFT_COLD_TABLE", None)
    for bv in ctx.values():
        for _ in range(120 if n <= 65536 else 40):
            bv.seals_run()
    res = {k: [] for k in ctx}
    stp = {k: [] for k in ctx}
    for rd in range(5):
        for k, bv in ctx.items():
            for _ in range(10):
                bv.seals_run()
            bv.set_kernel_timing(1)
            bv.last_kernel_ms()
            t0 = time.perf_counter()
            steps = 30 if n <= 65536 else 12
            for _ in range(steps):
                bv.seals_run()
            stp[k].append((time.perf_counter() - t0) / steps)
            ms, c = bv.last_kernel_ms()
            res[k].append(ms / c)
    for k, bv in ctx.items():
        m = float(np.median(res[k]))
        print(f"{n:7d}  {bv.last_dispatch()[0]:3d}  {k:9s} {m:.4f}   {m * 1e6 / n:6.2f}   {n / float(np.median(stp[k])) / 1e6:7.2f}")
        bv.close()
