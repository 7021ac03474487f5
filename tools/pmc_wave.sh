#!/bin/bash
# tools/pmc_wave.sh — instruction-mix counters of the dominant kernel (run through gpurun)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wave
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set -d "$OUT/$tag" -o p --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/$tag.log" 2>&1
done
cd "$ROOT"
python - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc_wave")
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (tot, n) in sorted(acc.items()):
    if "recover" in kn or "verify_known" in kn:
        print(f"{kn:42s} {c:24s} {tot / n:16.1f} per launch")
PY
