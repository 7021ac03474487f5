#!/bin/bash
# tools/trace_cmd.sh TAG -- <command…> : HIP API + kernel + memory-copy trace (no counters) of any command.
set -u
TAG=${1:-r02}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --stats -d "$OUT" -o q --output-format csv -- "$@" > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
for f in kernel_stats hip_api_stats memory_copy_stats; do [ -f "$OUT/q_$f.csv" ] && cp "$OUT/q_$f.csv" "$SUM/${TAG}_$f.csv"; done
cat "$SUM/${TAG}_kernel_stats.csv"
