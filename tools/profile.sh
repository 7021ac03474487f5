#!/bin/bash
# tools/profile.sh TAG [ROWS] — run on the GPU box (through gpurun): collects the rocprofv3 evidence
# bench.py's roofline object refers to, into gpurun_out/prof_TAG/ and summaries into
# gpurun_out/profiles/ (copy those into profiles/ and commit).
#   pass 1: --kernel-trace --stats           (per-kernel average duration; includes the config-#3 sequence kernels)
#   pass 2: --pmc FETCH_SIZE                 (own run: counters never share a run with traces)
#   pass 3: --pmc WRITE_SIZE
# LEAN=1 (bench.py's default run, round 6: the counters of the driver's own line come from the driver's own box): the
# headline stats pass and the two traffic passes only, shorter legs, 90 s per pass at most.
set -u
TAG=${1:-r03}
ROWS=${2:-4096}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
# the headline leg only: the sequence legs launch the same kernel over 8 192 rows (a COMMIT set), which would mix two
# launch shapes into one per-kernel average; their kernels are summarised by a second stats pass (…_seq_kernel_stats.csv)
LEAN=${LEAN:-0}
COMMON="--no-live-counters --no-cpu-baseline --no-sweep --no-sustained --no-certificates --no-host-mirror --extended-steps 0"
CMD="python $ROOT/bench.py --rows $ROWS --steps 50 --warmup 5 --no-sequence $COMMON"
SEQ="python $ROOT/bench.py --rows $ROWS --steps 10 --warmup 2 --no-warm --seq-rounds 30 $COMMON"
PMC="python $ROOT/bench.py --rows $ROWS --steps 30 --warmup 3 --no-sequence $COMMON"
T=600
if [ "$LEAN" = 1 ]; then
  T=90
  PMC="python $ROOT/bench.py --rows $ROWS --steps 20 --warmup 3 --no-sequence --no-warm $COMMON"
fi
cd /tmp
timeout $T rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
[ "$LEAN" = 1 ] || timeout $T rocprofv3 --kernel-trace --stats -d "$OUT/stats_seq" -o q --output-format csv -- $SEQ > "$OUT/stats_seq.log" 2>&1
timeout $T rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $PMC > "$OUT/fetch.log" 2>&1
timeout $T rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $PMC > "$OUT/write.log" 2>&1
cd "$ROOT"
python tools/summarize_prof.py "$OUT" "$SUM" "$TAG" "$ROWS"
f=$(find "$OUT/stats_seq" -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$SUM/${TAG}_seq_kernel_stats.csv"
[ -e "$SUM/${TAG}_kernel_stats.csv" ] && [ -e "$SUM/${TAG}_traffic.json" ]   # the exit code: did the passes bench.py reads leave their summaries
