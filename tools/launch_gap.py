"""Where do the ~29 us between the verdict kernel's duration and a step's wall time go?  One pass per host call
(what bench.py times) with and without the HIP-event pair around the kernel (IBFT_NO_EVENTS), and ten back-to-back
passes per call (ibft_seals_launch(repeat=10)).  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import go_ibft_amd.verifier as V
g = np.load("tests/golden/bench_round_n4096.npz")


def run(no_events):
    os.environ["IBFT_NO_EVENTS"] = "1" if no_events else "0"
    bv = V.BatchVerifier(max_rows=4096)
    bv.set_validators(1, g["addrs"], g["power"])
    bv.seals_stage(g["hash32"], g["seal65"], g["signer20"])
    for _ in range(30):
        bv.seals_run()
    bv.last_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(300):
        bv.seals_run()
    wall = (time.perf_counter() - t0) / 300 * 1e3
    ms, k = bv.last_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(30):
        bv.seals_launch(10); bv.seals_fetch()
    wall10 = (time.perf_counter() - t0) / 300 * 1e3
    ms10, k10 = bv.last_kernel_ms()
    print(f"events {'off' if no_events else 'on '}: one pass per call {wall:.4f} ms wall" + (f", kernel {ms / k:.4f} ms" if k else "") +
          f"; ten passes per call {wall10:.4f} ms wall per pass" + (f", kernel {ms10 / k10:.4f} ms" if k10 else ""))
    bv.close()


for ne in (False, True, False, True):
    run(ne)
