#!/bin/bash
# product verdict kernel vs the bare devtest kernel: same recover function — do they execute the same instructions?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_cmp
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OUT/dev" -o p --output-format csv -- python $ROOT/tools/rows_stages.py 4096 > "$OUT/dev.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OUT/prod" -o p --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sequence --no-warm > "$OUT/prod.log" 2>&1
cd "$ROOT"
python - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc_cmp")
for side in ("dev", "prod"):
    acc = {}
    for f in glob.glob(os.path.join(out, side, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"].split("(")[0][-48:]
            if "rows" not in kn: continue
            a = acc.setdefault((kn, row["Counter_Name"]), [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
    for (kn, c), (t, n) in sorted(acc.items()):
        print(f"{side:5s} {kn:50s} {c:18s} {t / n:16.1f} per launch ({n})")
PY
