#!/usr/bin/env python
"""tools/trace_timeline.py KERNEL_TRACE.csv ANCHOR [OCCURRENCE] [BEFORE] [AFTER] — print the kernels around the n-th dispatch
of the kernel whose name contains ANCHOR, with start / end relative to the first one shown (µs) and the queue they ran on:
shows what overlapped and where a stream waited (rocprofv3 --kernel-trace output)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 0
before = int(sys.argv[4]) if len(sys.argv) > 4 else 12
after = int(sys.argv[5]) if len(sys.argv) > 5 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i = idx[occ]
sel = rows[max(0, i - before): i + after]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name'][:58]:58s} q{r['Queue_Id']} start={s / 1e3:9.1f} end={e / 1e3:9.1f} dur={(e - s) / 1e3:8.1f} grid={r['Grid_Size_X']}")
