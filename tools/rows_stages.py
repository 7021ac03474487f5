"""Timing breakdown of the row-per-signature recover (sixteen lanes per signature, devtest build) at the
product's launch shape; run on the GPU box.  Usage: python tools/rows_stages.py [n_rows=4096]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (works from any directory: rocprofv3 runs it from /tmp)
import go_ibft_amd.build as build
from oracle import binding as B, workload as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = C.CDLL(os.environ.get("DEVTEST_SO") or build.build_devtest())
r = W.make_round(n, 5)
ms = (C.c_float * 9)()
out = np.zeros((n, 24), np.uint8)
rc = L.devtest_rows_stage_ms(n, r.hash32.tobytes(), r.seal65.tobytes(), ms, out.ctypes.data_as(C.c_void_p))
assert rc == 0
assert (out[:, 20] == 1).all() and (out[:, :20] == r.addrs).all(), "devtest rows kernel: wrong addresses"
names = ["sqrt + y", "+scalars (r^-1, GLV)", "+tables (T, TX)", "+main loop (128 dbl, 66 add)", "+16 G additions", "+Z^-1",
         "complete (+keccak)"]
prev = 0.0
print(f"# recover_pubkey_row, {n} rows = {(n + 3) // 4} wavefronts")
print(f"#   inside the scalar stage: after r^-1 mod n {ms[7]:.3f} ms, after u1 = -z/r, u2 = s/r {ms[8]:.3f} ms, after the GLV split {ms[1]:.3f} ms")
for nm, m in zip(names, list(ms)[:7]):
    print(f"{nm:32s} {m:7.3f} ms   (+{m - prev:.3f})")
    prev = m

# instruction-fetch share: the same wavefronts run the recover 1, 2, 3 times in ONE launch
ms3 = (C.c_float * 3)()
rc = L.devtest_rows_repeat_ms(n, r.hash32.tobytes(), r.seal65.tobytes(), 3, ms3)
assert rc == 0
print(f"# whole recover repeated inside one launch: 1x {ms3[0]:.3f} ms, 2x {ms3[1]:.3f} ms, 3x {ms3[2]:.3f} ms  ->  "
      f"a pass with the code already in the instruction cache: {ms3[2] - ms3[1]:.3f} ms; first pass costs {ms3[0] - (ms3[2] - ms3[1]):+.3f} ms more")
