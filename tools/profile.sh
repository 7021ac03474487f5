#!/bin/bash
# tools/profile.sh TAG — run on the GPU box (through gpurun): collects the rocprofv3 evidence
# bench.py's roofline object refers to, into gpurun_out/prof_TAG/ and summaries into
# gpurun_out/profiles/ (copy those into profiles/ and commit).
#   pass 1: --kernel-trace --stats           (per-kernel average duration)
#   pass 2: --pmc FETCH_SIZE                 (own run: counters never share a run with traces)
#   pass 3: --pmc WRITE_SIZE
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $CMD > "$OUT/write.log" 2>&1
cd "$ROOT"
python tools/summarize_prof.py "$OUT" "$SUM" "$TAG"
