"""Summarise tools/profile.sh output: kernel stats CSV + per-launch HBM traffic JSON.

Traffic follows /opt/skills/guides/MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE
are reported in KB (×1024 → bytes), collected in separate --pmc passes.  FETCH_SIZE is NOT
doubled here: the guide's gfx950 ×2 correction applies to wide coalesced streams (64-B
requests counted as 32-B); these kernels issue 16-B table/row loads, so the raw figure is the
honest one (doubling would only raise the number further above the algorithmic bytes).
"""
import csv
import glob
import json
import os
import shutil
import sys


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def per_kernel(path, counter):
    tot, cnt = {}, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return {k: tot[k] / cnt[k] for k in tot}


def main(out, summ, tag, rows=4096):
    stats = find(os.path.join(out, "stats"), "*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(summ, f"{tag}_kernel_stats.csv"))
    f = find(os.path.join(out, "fetch"), "*counter_collection.csv")
    w = find(os.path.join(out, "write"), "*counter_collection.csv")
    traffic = {}
    if f and w:
        fe, wr = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
        for k in fe:
            if "recover" in k or "verify_known" in k:
                traffic[k] = {
                    "rows": int(rows),
                    "fetch_kb": fe[k],
                    "write_kb": wr.get(k, 0.0),
                    "hbm_bytes_per_launch": int((fe[k] + wr.get(k, 0.0)) * 1024),
                    "note": "rocprofv3 FETCH_SIZE+WRITE_SIZE (KB) x 1024, separate passes, mean over launches; "
                            "FETCH not doubled (16-B table/row loads, not wide coalesced streams)",
                }
        with open(os.path.join(summ, f"{tag}_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    print(json.dumps({"stats": stats, "traffic": traffic}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
