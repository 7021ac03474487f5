#!/usr/bin/env python3
"""tools/asymptote.py N... — cold kernel time of one resident COMMIT batch of N seals (N beyond the sweep: 131 072, 262 144),
behind 40 untimed passes; IBFT_GPU_LIB picks the build.  The chip's asymptotic cold rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_ibft_amd.verifier as V
import go_ibft_amd.simulate as SIM

sizes = [int(x) for x in sys.argv[1:]] or [131072, 262144]
bv = V.BatchVerifier(flags=0, max_rows=max(sizes))
try:
    for n in sizes:
        r = SIM.make_round(bv, n, 100 + n)
        bv.set_validators(1, r.addrs, r.power)
        bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
        for _ in range(40):
            verdict, t = bv.seals_run()
        assert verdict.all() and t.has_quorum == 1
        bv.set_kernel_timing(1)
        bv.last_kernel_ms()
        for _ in range(30):
            bv.seals_run()
        k, l = bv.last_kernel_ms()
        ms = k / max(l, 1)
        print(f"{os.environ.get('IBFT_GPU_LIB', 'in-tree')} N={n}: kernel {ms:.4f} ms, {ms * 1e6 / n:.2f} ns per verify, {n / ms / 1e3:.1f} M verifies/s, dispatch {bv.last_dispatch()}", flush=True)
finally:
    bv.close()
