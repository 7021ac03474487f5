#!/usr/bin/env python3
"""bench.py's cpu_baseline leg alone (no GPU, no torch): the tuned CPU recovery and the plain checker path on this host's granted
cores over the N = 4 096 COMMIT batch.  usage: cpu_baseline_only.py [seconds] > out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import workload as W  # noqa: E402

rd = W.make_round(4096)
out = bench.cpu_baseline(rd.addrs, rd.power, rd.hash32, rd.seal65, rd.signer20, budget_s=float(sys.argv[1]) if len(sys.argv) > 1 else 6.0)
out["host"] = {"cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
               "usable_cores": bench.usable_cores()}
print(json.dumps(out, indent=1))
