#!/bin/bash
# tools/gpu_session_final.sh TAG — the round's closing evidence in ONE gpurun call: the whole GPU suite, the full soak, the bench
# line, rocprofv3 kernel stats + HBM traffic + instruction counters at N = 4 096, and the same at the other dispatch sizes.
set -u
TAG=${1:-r04k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
bash tools/gpu_session.sh $TAG tests stages bench prof pmc
timeout 1200 python tools/soak.py > gpurun_out/profiles/${TAG}_soak.json 2> gpurun_out/${TAG}_soak.err
echo "soak rc=$?"; tail -c 400 gpurun_out/profiles/${TAG}_soak.json; tail -2 gpurun_out/${TAG}_soak.err
timeout 1500 bash tools/profile_sizes.sh $TAG ${IBFT_PROF_SIZES:-512 1024 16384 65536} > gpurun_out/${TAG}_profile_sizes.log 2>&1
echo "profsizes rc=$?"; tail -5 gpurun_out/${TAG}_profile_sizes.log
