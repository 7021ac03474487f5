#!/usr/bin/env python3
"""tools/harness_ab.py MODE — the N = 4 096 cold kernel in ONE fresh process under one of three harness conditions (round 5: on one
box bench.py measured 0.3453 ms where tools/two_waves_ab.py measured 0.3335 ms for the same kernel on the same inputs):
  none        libibftgpu.so alone (what the A/B tools do)
  torch_first import torch, torch.cuda.set_device(0), THEN load libibftgpu.so (what bench.py did)
  lib_first   load libibftgpu.so (and create the context), THEN import torch + set_device
Prints the HIP runtime the process ended up with (which libamdhip64 is mapped) and the kernel time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd  # noqa: F401
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pin = None
if os.environ.get("IBFT_PIN") == "1":       # go_ibft_amd/numa.py: onto the GPU's own NUMA node, before any HIP call
    import go_ibft_amd.numa as NUMA
    pin = NUMA.pin_to_device_node(0)
if mode == "torch_first":
    import torch
    torch.cuda.set_device(0)
import numpy as np
import go_ibft_amd.verifier as V
V.load_library()
bv = V.BatchVerifier(flags=0, max_rows=4096)
if mode == "lib_first":
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
with np.load(os.path.join(ROOT, "tests", "golden", "bench_round_n4096.npz")) as z:
    g = {k: z[k] for k in z.files}
bv.set_validators(1, g["addrs"], g["power"])
bv.seals_stage(g["hash32"], g["seal65"], g["signer20"], None)
for _ in range(200):
    bv.seals_run()
bv.set_kernel_timing(1)
bv.last_kernel_ms()
t0 = time.perf_counter()
for _ in range(steps):
    verdict, t = bv.seals_run()
el = time.perf_counter() - t0
ms, k = bv.last_kernel_ms()
assert verdict.all() and t.has_quorum == 1
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
if pin is not None:
    print("pin:", pin, end="  ")
print(f"{mode:12s} kernel {ms / k:.4f} ms  step {el / steps * 1e3:.4f} ms  ({k} samples)  hip runtime: {', '.join(libs)}")
bv.close()
