#!/bin/bash
# tools/trace_seq.sh TAG — HIP API + kernel + memory-copy trace (no counters) of the config-#3 sequence:
# which runtime calls the per-call overhead of the five C-ABI calls is made of.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_seq
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --stats -d "$OUT" -o q --output-format csv -- \
  python $ROOT/tools/seq_breakdown.py 100 ${2:-} > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
find "$OUT" -name '*stats*.csv' | while read f; do b=$(basename "$f"); cp "$f" "$SUM/${TAG}_seq_${b#q_}"; done
ls -la "$SUM"; for f in "$SUM"/${TAG}_seq_*hip*stats*.csv "$SUM"/${TAG}_seq_*memory*stats*.csv; do echo "== $f"; head -25 "$f"; done
tail -5 "$OUT/run.log"
