#!/usr/bin/env python3
"""tools/side_tally_steps.py [N=4096] [warm=1] — per-pass view of a SHORT pipelined leg (the bench's K = 20): interval between
delivered passes and the verdict kernel's HIP-event time of every pass, IBFT_SIDE_TALLY as set in the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_ibft_amd.numa as NUMA
NUMA.pin_to_device_node(0)
import numpy as np
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if warm else 0, max_rows=max(n, 1024))
r = SIM.make_round(bv, n, 600 + n)
bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
for _ in range(150): bv.seals_run()
def leg(k, every):
    bv.set_kernel_timing(every); bv.last_kernel_ms()
    iv = []
    bv.seals_submit(); s0 = time.perf_counter()
    for _ in range(k - 1):
        bv.seals_submit(); bv.seals_collect()
        s1 = time.perf_counter(); iv.append((s1 - s0) * 1e6); s0 = s1
    bv.seals_collect(); iv.append((time.perf_counter() - s0) * 1e6)
    ms, cnt = bv.last_kernel_ms()
    return iv, ms / max(cnt, 1) * 1e3
leg(5, 0)
bv.sync()
for every in (4, 1, 0):
    iv, kus = leg(20, every)
    print(f"side={os.environ.get('IBFT_SIDE_TALLY', 'auto')} n={n} warm={warm} timing every {every}: step {sum(iv) / len(iv):.1f} us, kernel {kus:.1f} us; intervals",
          " ".join(f"{x:.0f}" for x in iv), f"| side tallies {bv.pipeline_stats()[0]}")
    bv.sync()
bv.close()
