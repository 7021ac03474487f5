#!/bin/bash
# tools/lease_pmc.sh TAG — instruction-cache / scalar-cache / wait counters of the lane kernel (N = 65 536) on this lease, next to its kind
# (tools/table_ab.py) and the instruction-fetch probe: what the leases of the slow kind do differently
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out/profiles
OUT=$ROOT/gpurun_out/lease_pmc_$TAG
mkdir -p "$OUT"
{ timeout 60 tools/scatter_probe | grep -E "loop body +(16|48|64|96) KB +wavefronts +1024"; timeout 120 python tools/table_ab.py 16384 65536 2>/dev/null | grep -E "lds"; } > gpurun_out/profiles/${TAG}_lease_pmc.txt 2>&1
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --rows 65536 --steps 12 --warmup 2 --no-cpu-baseline --no-sequence --no-sweep --no-sustained --no-certificates --no-host-mirror --no-warm --extended-steps 0"
cd /tmp
rocprofv3 -L > "$OUT/list.txt" 2>&1
for set in "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ" "SQ_IFETCH_LEVEL" "SQC_ICACHE_MISSES_DUPLICATE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  t=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set -d "$OUT/$t" -o p --output-format csv -- $CMD > "$OUT/$t.log" 2>&1
done
cd "$ROOT"
OUTDIR=$OUT python3 - >> gpurun_out/profiles/${TAG}_lease_pmc.txt <<'PY'
import csv, glob, os
acc = {}
for f in glob.glob(os.path.join(os.environ["OUTDIR"], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ecrecover_lane_kernel" not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(row["Counter_Name"], [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
for c, (tot, n) in sorted(acc.items()):
    print(f"ecrecover_lane_kernel<0, 0>  {c:28s} {tot / n:18.1f} per launch ({n} launches)")
PY
grep -i -E "icache|ifetch|SQC_TC" "$OUT/list.txt" | head -20 >> gpurun_out/profiles/${TAG}_lease_pmc.txt
cat gpurun_out/profiles/${TAG}_lease_pmc.txt
