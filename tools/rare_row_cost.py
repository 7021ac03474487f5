#!/usr/bin/env python
"""tools/rare_row_cost.py — what ONE crafted row costs a whole launch (a launch lasts as long as its slowest wavefront).
An honest COMMIT batch of N rows, then the same batch with one row replaced by a signature whose recover meets an
exceptional case in its last addition (u1·G = u2·R: z = s·k; u1·G = −u2·R: z = −s·k; u1 = 0: z = 0) — any validator can
send these.  Prints the verdict kernel's ms per variant (HIP events, median of 40 passes).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import go_ibft_amd.verifier as V
from oracle import pyref, workload as W

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
r = W.make_round(n_rows, 7)
N = pyref.N
rng = np.random.default_rng(3)


def crafted(kind):
    k = int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1
    x, y = pyref.pt_mul(k, pyref.G)
    rr, s = x % N, int.from_bytes(rng.bytes(32), "big") % (N - 1) + 1
    z = {"same": s * k % N, "opposite": (-s * k) % N, "zero_digest": 0}[kind]
    return z.to_bytes(32, "big"), rr.to_bytes(32, "big") + s.to_bytes(32, "big") + bytes([y & 1])


def kernel_ms(hash32, seal65):
    bv = V.BatchVerifier(max_rows=n_rows)
    try:
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(hash32, seal65, r.signer20, None)
        for _ in range(10):
            bv.seals_run()
        bv.last_kernel_ms()
        ms = []
        for _ in range(40):
            bv.seals_run()
            t, c = bv.last_kernel_ms()
            ms.append(t / max(c, 1))
        return float(np.median(ms))
    finally:
        bv.close()


base = kernel_ms(r.hash32, r.seal65)
print(f"honest batch of {n_rows} rows: {base:.4f} ms")
for kind in ("same", "opposite", "zero_digest"):
    h, sg = r.hash32.copy(), r.seal65.copy()
    z, sig = crafted(kind)
    h[17] = np.frombuffer(z, dtype=np.uint8)
    sg[17] = np.frombuffer(sig, dtype=np.uint8)
    ms = kernel_ms(h, sg)
    print(f"one row with {kind:12s}: {ms:.4f} ms  ({(ms / base - 1) * 100:+.1f} %)")
