"""Timing breakdown of the one-wavefront-per-signature recover (devtest build, run on the GPU box)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import go_ibft_amd.build as build
from oracle import workload as W
L = C.CDLL(os.environ.get("DEVTEST_SO") or build.build_devtest())
r = W.make_round(1024, 5)
ms = (C.c_float * 7)()
rc = L.devtest_wave_stage_ms(1024, r.hash32.tobytes(), r.seal65.tobytes(), ms)
assert rc == 0
names = ["prefix+sqrt", "+scalars (r^-1, GLV)", "+table", "+main loop", "+y fix, G part", "+combine, Z^-1", "complete (+keccak)"]
prev = 0.0
for n, m in zip(names, ms):
    print(f"{n:24s} {m:7.3f} ms   (+{m - prev:.3f})")
    prev = m
