#!/usr/bin/env python3
"""tools/warm_ab.py OLD.so NEW.so — the warm kernels of two builds on ONE lease, separate processes alternating (IBFT_GPU_LIB),
next to the lease's kind (cold group-4 kernel at 16 384 rows: ≈ 0.70 ms fast kind, ≈ 0.85 ms slow kind)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import go_ibft_amd.numa as NUMA
NUMA.pin_to_device_node(0)
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
out = {}
for n, flags in ((16384, 0), (65536, 0), (40000, 0)):
    bv = V.BatchVerifier(flags=flags, max_rows=n)
    r = SIM.make_round(bv, n, 600 + n)
    bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
    for _ in range(150): v, t = bv.seals_run()
    assert v.all() and t.has_quorum == 1
    bv.set_kernel_timing(1); bv.last_kernel_ms()
    for _ in range(60): bv.seals_run()
    ms, k = bv.last_kernel_ms()
    out[("warm" if flags else "cold") + str(n)] = round(ms / k, 4)
    bv.close()
print(json.dumps(out))
''' % ROOT
old, new = sys.argv[1], sys.argv[2]
for rd in range(3):
    for name, lib in (("old", old), ("new", new)):
        env = dict(os.environ, IBFT_GPU_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(name, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
