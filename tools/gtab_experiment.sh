#!/bin/bash
# the 84 MB fixed-base table for G (16 windows x 65 536 entries, 16 mixed additions per signature)
# against an L2-resident one (32 windows x 256 entries = 655 KB, 32 mixed additions): time and HBM fetch traffic.
# Needs go-ibft_amd/csrc/libibftgpu_g8.so (same sources, -DIBFT_GTAB_BITS=8).  Run through gpurun.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gtab
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles"
export TMPDIR=/tmp
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=r.get("warm_path") or {}; print("  rows", r["config"]["rows_per_gpu"], "cold", r["roofline"]["kernel"], "kernel_ms %.4f" % r["roofline"]["avg_kernel_ms"], "verifies/s %d" % r["value"], "| warm", w.get("kernel"), "kernel_ms %.4f" % w.get("kernel_ms", 0), "verifies/s %d" % w.get("value", 0))'
{
for lib in "" go-ibft_amd/csrc/libibftgpu_g8.so; do
  echo "== G table: ${lib:-16-bit windows, 84 MB (product)}"
  for n in 1024 4096; do
    IBFT_GPU_LIB=$lib python $ROOT/bench.py --rows $n --steps 100 --warmup 10 --no-cpu-baseline --no-sequence | python -c "$P"
  done
  cd /tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/f${lib:+8}" -o f --output-format csv -- env IBFT_GPU_LIB=$lib python $ROOT/bench.py --rows 4096 --steps 20 --warmup 3 --no-cpu-baseline --no-sequence > "$OUT/f.log" 2>&1
  cd "$ROOT"
  python - "$OUT/f${lib:+8}" <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"].split("(")[0][-44:]
        if "recover" in kn or "verify_known" in kn:
            a = acc.setdefault(kn, [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
for kn, (t, n) in sorted(acc.items()):
    print(f"  FETCH_SIZE at 4096 rows  {kn:46s} {t / n * 1024 / 1e6:8.2f} MB per launch")
PY
done
} | tee $ROOT/gpurun_out/profiles/r02_gtab_experiment.txt
