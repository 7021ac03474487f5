#!/usr/bin/env python3
"""tools/pair_ab.py — one wavefront per signature against TWO (main + helper, round 4) at the batch sizes where at most half
of the chip's SIMDs would otherwise work: kernel ms by HIP events, 60 synchronous passes per case, seals (MODE 0) and
senders (MODE 1).   python tools/pair_ab.py > gpurun_out/profiles/r04_pair_ab.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import occupancy as O  # noqa: E402


def main():
    import go_ibft_amd.verifier as V
    import go_ibft_amd.simulate as SIM
    print("# tools/pair_ab.py — cold path, rows <= 1024: IBFT_COLD_LANES = 64 (one wavefront per signature) against 128 (two)")
    print(f"{'rows':>6s} {'one wavefront: kernel ms':>26s} {'ms/step':>8s} {'two wavefronts: kernel ms':>26s} {'ms/step':>8s} {'gain':>6s}")
    for n in (1, 64, 128, 256, 384, 512, 768, 1024):
        a = O.one(V, SIM, "cold", n, 64, 60)
        b = O.one(V, SIM, "cold", n, 128, 60)
        print(f"{n:6d} {a['kernel_ms']:26.4f} {a['ms_per_step']:8.4f} {b['kernel_ms']:26.4f} {b['ms_per_step']:8.4f} "
              f"{(1 - b['kernel_ms'] / a['kernel_ms']) * 100:5.1f}%", flush=True)


if __name__ == "__main__":
    main()
