#!/usr/bin/env python3
"""tools/rows_stage_issue.py [rows=4096] — WHERE the row-per-signature recover (the headline kernel's body) waits: per stage,
the time (tools/rows_stages.py: devtest kernels cut short after each stage, HIP events) next to the VALU instructions the stage
issues (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over the same command, a run of its own) → ns per wave-instruction per SIMD.  With
one resident wavefront per SIMD an instruction issues in 1.79 ns (profiles/r05a_ubench_wave.txt); a stage above that waits for
its own results (dependency stalls in a single chain), and the excess × its instruction count is all that interleaving it with
independent work could ever recover (round-5 review, item 4: "or the A/B that shows why not").  Run on the GPU box."""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "4096"
env = dict(os.environ, TMPDIR="/tmp")
times = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rows_stages.py"), n], cwd=ROOT, capture_output=True, text=True, env=env)
print(times.stdout, end="")
stage_ms = {}
for line in times.stdout.splitlines():
    m = re.match(r"^(\S.*?)\s+([0-9.]+) ms\s+\(\+", line)
    if m:
        stage_ms[m.group(1).strip()] = float(m.group(2))
m = re.search(r"after r\^-1 mod n ([0-9.]+) ms, after u1 = -z/r, u2 = s/r ([0-9.]+) ms", times.stdout)
sub = (float(m.group(1)), float(m.group(2))) if m else (None, None)
d = tempfile.mkdtemp(prefix="rows_issue_", dir="/tmp")
rp = subprocess.run(["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "-d", d, "-o", "c", "--output-format", "csv", "--",
                     sys.executable, os.path.join(ROOT, "tools", "rows_stages.py"), n], cwd="/tmp", capture_output=True, text=True, env=env)
if rp.returncode != 0:
    print("rocprofv3 returned", rp.returncode, rp.stderr[-1500:], file=sys.stderr)
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.search(r"devtest_rows_recover_kernel<(\d+)>", row["Kernel_Name"])
        if k and row["Counter_Name"] == "SQ_INSTS_VALU":
            a = acc.setdefault(int(k.group(1)), [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
insts = {k: v[0] / v[1] for k, v in acc.items()}
order = [(1, "t = x^3+7, R'", "sqrt + y"), (21, "r^-1 mod n (safegcd)", None), (22, "u1, u2", None), (2, "GLV split, digits", "+scalars (r^-1, GLV)"),
         (3, "window tables", "+tables (T, TX)"), (4, "main loop", "+main loop (128 dbl, 66 add)"), (5, "16 G additions", "+16 G additions"),
         (6, "join + closing exponentiation", "+Z^-1"), (99, "Keccak, compare", "complete (+keccak)")]
ms_at = {1: stage_ms.get("sqrt + y"), 21: sub[0], 22: sub[1], 2: stage_ms.get("+scalars (r^-1, GLV)"), 3: stage_ms.get("+tables (T, TX)"),
         4: stage_ms.get("+main loop (128 dbl, 66 add)"), 5: stage_ms.get("+16 G additions"), 6: stage_ms.get("+Z^-1"), 99: stage_ms.get("complete (+keccak)")}
waves = (int(n) + 3) // 4
print(f"# per stage: ms, VALU wave-instructions per wavefront (SQ_INSTS_VALU / {waves} wavefronts), ns per instruction per SIMD; excess over 1.79 ns x instructions = stall")
prev_ms, prev_i, tot_stall = 0.0, 0.0, 0.0
for key, name, _ in order:
    if key not in insts or ms_at.get(key) is None:
        continue
    dms, di = ms_at[key] - prev_ms, (insts[key] - prev_i) / waves
    ns = dms * 1e6 / di if di > 0 else float("nan")
    stall = max(0.0, dms * 1e3 - di * 1.79e-3)
    tot_stall += stall
    print(f"{name:34s} {dms:7.4f} ms  {di:9.0f} inst  {ns:6.2f} ns/inst   stall {stall:6.1f} us")
    prev_ms, prev_i = ms_at[key], insts[key]
print(f"# total {prev_ms:.4f} ms, {prev_i / waves:.0f} instructions per wavefront = {prev_i / waves * 1.79e-3:.1f} us at 1.79 ns; stalls {tot_stall:.1f} us")
