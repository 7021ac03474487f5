#!/usr/bin/env python3
"""tools/two_waves_ab.py — what a second resident wavefront per SIMD buys the row-per-signature recover, measured on ONE box
in ONE process (round-4 review, "next round" #3: a two-wavefronts-per-SIMD kernel for N = 4 096, "or a same-box A/B that
shows why not").

ecrecover_rows_kernel<0> puts rows × 16 / 64 wavefronts on 1 024 SIMDs: 4 096 rows = one wavefront per SIMD, 8 192 rows =
two (174 registers: two fit).  The SAME code, the SAME instruction mix; the only difference is that every SIMD has a second
instruction stream to issue from.  ns per verify at 8 192 rows ÷ ns per verify at 4 096 rows is therefore the issue-rate
gain any two-wavefront design of this arithmetic can hope for — a design that splits a signature over 32 lanes must cost
LESS than that factor in additional instructions to win.

Passes alternate between the two sizes (two contexts), each behind ≥ 150 untimed passes; kernel time by HIP events.

    python tools/two_waves_ab.py [--rounds 6] [--steps 40] > gpurun_out/profiles/r05_two_waves_ab.txt"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    os.environ["IBFT_COLD_LANES"] = "16"          # pin the row-per-signature kernel at both sizes
    import go_ibft_amd.verifier as V
    import go_ibft_amd.simulate as SIM
    ctx = {}
    for n in (4096, 8192, 16384):
        bv = V.BatchVerifier(flags=0, max_rows=n)
        r = SIM.make_round(bv, n, 500 + n)
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
        verdict, t = bv.seals_run()
        assert verdict.all() and t.has_quorum == 1 and bv.last_dispatch()[0] == 16
        ctx[n] = bv
    for n, bv in ctx.items():
        for _ in range(150):
            bv.seals_run()
    res = {n: [] for n in ctx}
    for rd in range(args.rounds):
        for n, bv in ctx.items():
            for _ in range(20):
                bv.seals_run()
            bv.set_kernel_timing(1)
            bv.last_kernel_ms()
            for _ in range(args.steps):
                bv.seals_run()
            ms, k = bv.last_kernel_ms()
            res[n].append(ms / k)
    print("# ecrecover_rows_kernel<0> (IBFT_COLD_LANES=16), kernel ms by HIP events, mean of", args.steps, "passes per round; rounds alternate between the sizes")
    print("# rows  wavefronts/SIMD offered (2 resident at most)  kernel ms per round ...  median  ns/verify")
    med = {}
    for n in ctx:
        m = float(np.median(res[n]))
        med[n] = m
        print(f"{n:6d}  {n * 16 / 64 / 1024:4.1f}  " + " ".join(f"{x:.4f}" for x in res[n]) + f"   {m:.4f}   {m * 1e6 / n:.2f}")
    g2 = (med[4096] / 4096) / (med[8192] / 8192)
    g4 = (med[4096] / 4096) / (med[16384] / 16384)
    print(f"# issue-rate gain of a second resident wavefront, same code: {g2:.3f}x   (4 offered / 2 resident: {g4:.3f}x)")
    print(f"# a 32-lanes-per-signature design wins at N = 4 096 only if it costs less than {g2:.3f}x the instructions per signature")
    for bv in ctx.values():
        bv.close()


if __name__ == "__main__":
    main()
