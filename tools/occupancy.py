#!/usr/bin/env python3
"""tools/occupancy.py — the occupancy experiment, run instead of argued (round-3 review, "next round" #5).

For every (rows, lanes-per-signature) pair: one resident COMMIT batch signed on the device, `steps` synchronous passes
(verdict kernel + tally, results host-visible), kernel time from HIP events around every pass.  Lanes per signature decide
how many wavefronts a batch puts on a SIMD (rows × lanes / 64 wavefronts over 1 024 SIMDs); the AUTO dispatch keeps the
cold path at one wavefront per SIMD at every size — this measures what happens when it does not.

    python tools/occupancy.py [--steps 30] [--quick]  > gpurun_out/profiles/r04_occupancy.txt

Also: N = 4 096 as two concurrent streams of 2 048-row one-wavefront-per-signature kernels (two contexts, two host
threads) against the one 4 096-row row-per-signature launch.
"""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VGPRS = {"cold": {1: 243, 2: 256 + 13, 4: 244, 8: 256 + 57, 16: 174, 64: 182},
         "warm": {1: 235, 2: 256 + 16, 4: 234, 8: 234, 16: 234, 32: 234, 64: 164}}


def resident_waves(path: str, lanes: int) -> int:
    """wavefronts one SIMD can hold for this variant (512 registers per lane per SIMD, VGPR + AGPR)"""
    return max(1, 512 // (((VGPRS[path][lanes] + 7) // 8) * 8))


def one(V, SIM, path: str, n: int, lanes: int | None, steps: int, distinct: int | None = None):
    env = "IBFT_COLD_LANES" if path == "cold" else "IBFT_WARM_LANES"
    os.environ.pop("IBFT_COLD_LANES", None)
    os.environ.pop("IBFT_WARM_LANES", None)
    if lanes:
        os.environ[env] = str(lanes)
    flags = V.FLAG_PUBKEY_CACHE if path == "warm" else 0
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    try:
        d = distinct or n
        r = SIM.make_round(bv, d, 100 + d)
        reps = n // d
        tile = (lambda a: np.tile(a, (reps, 1)) if a.ndim == 2 else np.tile(a, reps))
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(tile(r.hash32), tile(r.seal65), tile(r.signer20), None)
        for _ in range(3):
            verdict, t = bv.seals_run()
        assert verdict.all() and t.has_quorum == 1 and t.distinct_senders == d, (n, lanes, t.distinct_senders)
        bv.set_kernel_timing(1)
        bv.last_kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            bv.seals_run()
        el = time.perf_counter() - t0
        kms, kl = bv.last_kernel_ms()
        cold_l, warm_l = bv.last_dispatch()
        used = cold_l if path == "cold" else (bv.cache_stats() and bv.lanes_per_signature)
    finally:
        bv.close()
        os.environ.pop(env, None)
    waves = n * used / 64.0
    return {"path": path, "rows": n, "lanes": used, "forced": bool(lanes), "kernel_ms": kms / max(kl, 1), "ms_per_step": el / steps * 1e3,
            "mverifies_s": n * steps / el / 1e6, "waves_per_simd": waves / 1024.0,
            "resident_cap": resident_waves(path, used) if used in VGPRS[path] else None}


def two_streams(V, SIM, steps: int):
    """N = 4 096 as two contexts of 2 048 rows each, driven by two host threads at once (one-wavefront-per-signature kernels,
    2 × 2 048 wavefronts = 4 per SIMD offered, 2 resident) against one 4 096-row row-per-signature launch"""
    os.environ["IBFT_COLD_LANES"] = "64"
    ctxs = []
    try:
        seed_bv = V.BatchVerifier(max_rows=4096)
        r = SIM.make_round(seed_bv, 4096, 4196)
        seed_bv.close()
        for h in range(2):
            bv = V.BatchVerifier(max_rows=2048)
            bv.set_validators(1, r.addrs, r.power)
            s = slice(2048 * h, 2048 * (h + 1))
            bv.seals_stage(r.hash32[s], r.seal65[s], r.signer20[s], None)
            for _ in range(3):
                v, _ = bv.seals_run()
            assert v.all()
            ctxs.append(bv)
        barrier = threading.Barrier(3)

        def worker(bv):
            barrier.wait()
            for _ in range(steps):
                bv.seals_run()
            barrier.wait()
        th = [threading.Thread(target=worker, args=(bv,)) for bv in ctxs]
        for t in th:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        el = time.perf_counter() - t0
        for t in th:
            t.join()
    finally:
        for bv in ctxs:
            bv.close()
        os.environ.pop("IBFT_COLD_LANES", None)
    return {"rows": 4096, "form": "2 contexts x 2048 rows, ecrecover_wave_kernel, two host threads", "ms_per_4096_rows": el / steps * 1e3,
            "mverifies_s": 4096 * steps / el / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import go_ibft_amd.verifier as V
    import go_ibft_amd.simulate as SIM
    plan_cold = [(4096, None), (4096, 64), (8192, None), (8192, 8), (8192, 4), (16384, None), (16384, 8), (16384, 16), (16384, 2),
                 (32768, None), (32768, 4), (32768, 1), (65536, None), (65536, 2), (65536, 4), (131072, None), (131072, 2), (262144, None)]
    plan_warm = [(4096, None), (4096, 8), (4096, 32), (16384, None), (16384, 8), (16384, 2), (65536, None), (65536, 2), (65536, 4),
                 (131072, None), (131072, 2), (262144, None), (262144, 2)]
    if args.quick:
        plan_cold, plan_warm = plan_cold[:3], plan_warm[:2]
    print("# tools/occupancy.py — kernel ms by HIP events around every pass; M/s = rows x steps / wall time of the synchronous passes")
    print(f"# steps per case: {args.steps}; 'waves/SIMD' = rows x lanes / 64 / 1024 offered, 'cap' = wavefronts a SIMD can hold (registers)")
    print(f"{'path':5s} {'rows':>7s} {'lanes':>5s} {'auto':>4s} {'waves/SIMD':>10s} {'cap':>3s} {'kernel ms':>10s} {'ms/step':>8s} {'M verifies/s':>12s} {'ns/verify':>9s}")
    for path, plan in (("cold", plan_cold), ("warm", plan_warm)):
        for n, lanes in plan:
            try:
                # the key cache holds 655 KB per validator: beyond 65 536 distinct validators the batch repeats the set
                e = one(V, SIM, path, n, lanes, args.steps if n <= 65536 else max(8, args.steps // 3), distinct=min(n, 65536))
                print(f"{e['path']:5s} {e['rows']:7d} {e['lanes']:5d} {'no' if e['forced'] else 'yes':>4s} {e['waves_per_simd']:10.2f} "
                      f"{str(e['resident_cap']):>3s} {e['kernel_ms']:10.4f} {e['ms_per_step']:8.4f} {e['mverifies_s']:12.2f} "
                      f"{e['kernel_ms'] * 1e6 / e['rows']:9.2f}", flush=True)
            except Exception as ex:  # noqa: BLE001
                print(f"{path:5s} {n:7d} {str(lanes):>5s}  failed: {ex!r}", flush=True)
    try:
        t = two_streams(V, SIM, args.steps)
        print(f"# N = 4096 as {t['form']}: {t['ms_per_4096_rows']:.4f} ms per 4096 rows, {t['mverifies_s']:.2f} M verifies/s")
    except Exception as ex:  # noqa: BLE001
        print(f"# two streams failed: {ex!r}")


if __name__ == "__main__":
    main()
