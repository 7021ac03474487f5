#!/bin/bash
# tools/gpu_session_close.sh TAG — gpu_session_final.sh behind a gate: the GPU suite first, the rest only when it is green
# (a red suite must not spend the box time of the soak and the profiles).
set -u
TAG=${1:-r04s}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" gpurun_out/${TAG}_pytest_gpu.log | head -20; exit $rc; }
cp gpurun_out/${TAG}_pytest_gpu.log gpurun_out/profiles/
bash tools/gpu_session.sh $TAG stages bench prof pmc
timeout 1200 python tools/soak.py > gpurun_out/profiles/${TAG}_soak.json 2> gpurun_out/${TAG}_soak.err
echo "soak rc=$?"; tail -c 300 gpurun_out/profiles/${TAG}_soak.json
timeout 1500 bash tools/profile_sizes.sh $TAG ${IBFT_PROF_SIZES:-512 1024 16384 65536} > gpurun_out/${TAG}_profile_sizes.log 2>&1
echo "profsizes rc=$?"; tail -3 gpurun_out/${TAG}_profile_sizes.log
