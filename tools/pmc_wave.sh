#!/bin/bash
# tools/pmc_wave.sh [ROWS] [TAG] — instruction-mix counters of the dominant kernels (run through gpurun);
# writes gpurun_out/profiles/TAG_pmc_instruction_mix.txt (copy into profiles/ and commit)
set -u
ROWS=${1:-4096}
TAG=${2:-r02_n$ROWS}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wave_$TAG
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
# LEAN=1 (bench.py's default run): the instruction counts and the cycle counters only, cold kernel only, 90 s per pass at most
LEAN=${LEAN:-0}
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH" "GRBM_GUI_ACTIVE")
T=300; EXTRA=""
if [ "$LEAN" = 1 ]; then
  SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY")
  T=90; EXTRA="--no-warm"
fi
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_')
  timeout $T rocprofv3 --pmc $set -d "$OUT/$tag" -o p --output-format csv -- python $ROOT/bench.py --rows $ROWS --steps 20 --warmup 3 $EXTRA --no-live-counters --no-cpu-baseline --no-sequence --no-sweep --no-sustained --no-certificates --no-host-mirror --extended-steps 0 > "$OUT/$tag.log" 2>&1
done
cd "$ROOT"
ROWS=$ROWS OUTDIR=$OUT python - > "$SUM/${TAG}_pmc_instruction_mix.txt" <<'PY'
import csv, glob, os
out = os.environ["OUTDIR"]
print(f"# rows per launch: {os.environ['ROWS']}")
print("# rocprofv3 --pmc, one counter set per run of `bench.py --rows ROWS --steps 20 --warmup 3 --no-cpu-baseline --no-sequence`; mean over launches")
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][-44:], row["Counter_Name"])
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (tot, n) in sorted(acc.items()):
    if "recover" in kn or "verify_known" in kn or "tally" in kn:
        print(f"{kn:46s} {c:24s} {tot / n:16.1f} per launch   ({n} launches)")
PY
cat "$SUM/${TAG}_pmc_instruction_mix.txt"
