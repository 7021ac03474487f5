#!/usr/bin/env python3
"""tools/fresh_batch_ab.py OLD.so NEW.so [rounds] — the cold verdict kernel on a RESIDENT batch re-verified every pass (what
every bench leg does) against FRESH batches (a different round's seals every pass: what a node sees), for two builds of the
library on one lease, separate processes alternating.  A batch that repeats keeps its sixteen G-table entries per row in L2;
a fresh one reads them from the Infinity Cache or HBM (K batches × N rows × 16 entries × one 128-byte line each are made to
exceed the 256 MB Infinity Cache).  Round 6's prefetch of those entries (rows kernel: into LDS as soon as u1 is known; lane /
group kernels: one step ahead) is invisible on the resident batch and is what the fresh column measures.  Kernel ms by HIP
events around every pass.

    python tools/fresh_batch_ab.py ab/libibftgpu_r05.so go-ibft_amd/csrc/libibftgpu.so 3 > gpurun_out/profiles/r06b_fresh_batch_ab.txt"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import go_ibft_amd.numa as NUMA
NUMA.pin_to_device_node(0)
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
out = {}
for n, K in ((4096, 48), (16384, 12), (65536, 4)):
    bv = V.BatchVerifier(flags=0, max_rows=n)
    rounds = [SIM.make_round(bv, n, 7, round_=k) for k in range(K)]
    cols = [tuple(V.pinned_copy(x) for x in (r.hash32, r.seal65, r.signer20)) for r in rounds]
    bv.set_validators(1, rounds[0].addrs, rounds[0].power)
    bv.seals_stage(*cols[0])
    for _ in range(150): v, t = bv.seals_run()
    assert v.all() and t.has_quorum == 1
    bv.set_kernel_timing(1); bv.last_kernel_ms()
    for _ in range(60): bv.seals_run()
    ms, k = bv.last_kernel_ms()
    out["resident%%d" %% n] = round(ms / k, 5)
    for k in range(K):                      # one untimed lap: every batch verified once (and checked)
        bv.seals_stage(*cols[k]); v, t = bv.seals_run(); assert v.all() and t.has_quorum == 1
    bv.last_kernel_ms()
    laps = max(2, 96 // K)
    for _ in range(laps):
        for k in range(K):
            bv.seals_stage(*cols[k]); bv.seals_run()
    ms, k = bv.last_kernel_ms()
    out["fresh%%d" %% n] = round(ms / k, 5)
    bv.close()
print(json.dumps(out))
''' % ROOT
old, new = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
acc = {"old": {}, "new": {}}
for rd in range(rounds):
    for name, lib in (("old", old), ("new", new)):
        env = dict(os.environ, IBFT_GPU_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        print(name, line or p.stderr[-400:], flush=True)
        try:
            for k, v in json.loads(line).items():
                acc[name].setdefault(k, []).append(v)
        except ValueError:
            pass
print("# median kernel ms: what  old  new  new/old")
for k in sorted(acc["old"], key=lambda x: (int(x.lstrip("residentfh")), x)):
    if k in acc["new"]:
        a, b = float(np.median(acc["old"][k])), float(np.median(acc["new"][k]))
        print(f"# {k:>14s}  {a:.5f}  {b:.5f}  {b / a:.4f}")
