"""Does the verdict kernel run slower when a host round trip separates the launches?  One pass per host call
(what bench.py times) against ten back-to-back passes per call (ibft_seals_launch(repeat=10)).  GPU box only."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import go_ibft_amd.verifier as V
g = np.load("tests/golden/bench_round_n4096.npz")
bv = V.BatchVerifier(max_rows=4096)
bv.set_validators(1, g["addrs"], g["power"])
bv.seals_stage(g["hash32"], g["seal65"], g["signer20"])
for _ in range(20):
    bv.seals_run()
bv.last_kernel_ms()
for _ in range(100):
    bv.seals_run()
ms, k = bv.last_kernel_ms()
print(f"one pass per host call:      {ms / k:.4f} ms per verdict kernel ({k} launches)")
for _ in range(10):
    bv.seals_launch(10); bv.seals_fetch()
ms, k = bv.last_kernel_ms()
print(f"ten passes per host call:    {ms / k:.4f} ms per verdict kernel ({k} launches)")
bv.close()
