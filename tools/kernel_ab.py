#!/usr/bin/env python3
"""tools/kernel_ab.py OLD.so NEW.so [rounds] [N …] — the cold verdict kernels of two builds of libibftgpu.so on ONE lease,
separate processes alternating (IBFT_GPU_LIB), HIP-event kernel time behind 150 untimed passes at every size the AUTO rule
serves with a different kernel: 64 (two wavefronts per signature), 1 024 (one wavefront), 4 096 (rows), 16 384 (4-lane
groups), 32 768 (2-lane groups), 65 536 (lane); a size written wN is the WARM path at N rows (keys known).  Also prints the device canary of every process (ibft_issue_probe) so that a
slow device is not mistaken for a slow build, and — the lease's kind — the old build's 16 384-row time.

    python tools/kernel_ab.py ab/libibftgpu_r05.so go-ibft_amd/csrc/libibftgpu.so 3 > gpurun_out/profiles/r06a_kernel_ab.txt"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import go_ibft_amd.numa as NUMA
NUMA.pin_to_device_node(0)
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
sizes = [x for x in sys.argv[1:]]          # "4096" = cold, "w4096" = warm (keys known)
out = {}
for key in sizes:
    warm = key.startswith("w")
    n = int(key.lstrip("w"))
    bv = V.BatchVerifier(flags=V.FLAG_PUBKEY_CACHE if warm else 0, max_rows=max(n, 1024))
    r = SIM.make_round(bv, n, 600 + n)
    bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
    for _ in range(150): v, t = bv.seals_run()
    assert v.all() and t.has_quorum == 1
    if warm: assert bv.cache_stats()[0] == n
    bv.set_kernel_timing(1); bv.last_kernel_ms()
    for _ in range(60): bv.seals_run()
    ms, k = bv.last_kernel_ms()
    out[key] = round(ms / k, 5)
    if key == sizes[-1]:
        out["canary_ns"] = round(bv.issue_probe()[0], 4)
    bv.close()
print(json.dumps(out))
''' % ROOT
old, new = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sizes = sys.argv[4:] or ["64", "1024", "4096", "16384", "32768", "65536", "w1024", "w4096", "w16384", "w65536"]
acc = {"old": {}, "new": {}}
for rd in range(rounds):
    for name, lib in (("old", old), ("new", new)):
        env = dict(os.environ, IBFT_GPU_LIB=os.path.abspath(lib), IBFT_MIN_ABI="3")   # (an older build: without the exports that came later)
        p = subprocess.run([sys.executable, "-c", CHILD] + sizes, env=env, capture_output=True, text=True, timeout=600)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        print(name, line or p.stderr[-400:], flush=True)
        try:
            for k, v in json.loads(line).items():
                acc[name].setdefault(k, []).append(v)
        except ValueError:
            pass
print("# median kernel ms: N  old  new  new/old")
for k in sizes + ["canary_ns"]:
    if k in acc["old"] and k in acc["new"]:
        a, b = float(np.median(acc["old"][k])), float(np.median(acc["new"][k]))
        print(f"# {k:>9s}  {a:.5f}  {b:.5f}  {b / a:.4f}")
if "16384" in acc["old"]:
    m = float(np.median(acc["old"]["16384"]))
    print(f"# lease kind by the OLD build's 16 384-row kernel ({m:.3f} ms): {'fast' if m < 0.77 else 'slow'} (0.70 / 0.85 ms, DESIGN.md §5.8)")
