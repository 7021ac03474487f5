#!/bin/bash
# tools/profile_sizes.sh TAG ROWS... — rocprofv3 evidence for the batch sizes the N = 4 096 profiles do not cover
# (run through gpurun).  Per size: one --kernel-trace --stats pass (cold + warm legs of bench.py), FETCH_SIZE and
# WRITE_SIZE in passes of their own, and two instruction-mix counter sets; plus, once, the kernel stats of the
# N = 256 round change from the transport's bytes (the cert_* kernels).  Summaries: gpurun_out/profiles/TAG_n<ROWS>_*.
set -u
TAG=${1:-r04}
shift || true
SIZES=${*:-1024 16384 65536}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$SUM"
export TMPDIR=/tmp
for ROWS in $SIZES; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_n$ROWS
  mkdir -p "$OUT"
  CMD="python $ROOT/bench.py --rows $ROWS --steps 30 --warmup 3 --no-live-counters --no-cpu-baseline --no-sequence --no-sweep --no-sustained --no-certificates --no-host-mirror --extended-steps 0"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $CMD > "$OUT/write.log" 2>&1
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    t=$(echo $set | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $set -d "$OUT/pmc_$t" -o p --output-format csv -- $CMD > "$OUT/pmc_$t.log" 2>&1
  done
  cd "$ROOT"
  python tools/summarize_prof.py "$OUT" "$SUM" "${TAG}_n$ROWS" "$ROWS" > "$OUT/summary.log" 2>&1
  ROWS=$ROWS OUTDIR=$OUT python - > "$SUM/${TAG}_n${ROWS}_pmc_instruction_mix.txt" <<'PY'
import csv, glob, os
out = os.environ["OUTDIR"]
print(f"# rows per launch: {os.environ['ROWS']}")
print("# rocprofv3 --pmc, one counter set per run of `bench.py --rows ROWS --steps 30 --warmup 3 --no-cpu-baseline --no-sequence ...`; mean over launches")
acc = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][-44:], row["Counter_Name"])
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (tot, n) in sorted(acc.items()):
    if "recover" in kn or "verify_known" in kn or "tally" in kn:
        print(f"{kn:46s} {c:24s} {tot / n:16.1f} per launch   ({n} launches)")
PY
  echo "== $ROWS"; head -6 "$SUM/${TAG}_n${ROWS}_kernel_stats.csv" 2>/dev/null
done
# the certificate kernels of one N = 256 round change (29 412 signatures from the transport's bytes)
OUT=$ROOT/gpurun_out/prof_${TAG}_cert
mkdir -p "$OUT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o c --output-format csv -- python $ROOT/tools/cert_from_wire.py 256 > "$OUT/stats.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$SUM/${TAG}_cert_n256_kernel_stats.csv" && head -14 "$f"
