#!/bin/bash
# tools/gpu_session.sh TAG [what...] — one gpurun call: tests, microbenchmarks, the bench line and the rocprof
# evidence for it.  Everything lands under gpurun_out/ (summaries in gpurun_out/profiles/).
#   what: tests ubench stages bench prof pmc (default: all)
set -u
TAG=${1:-r03}
shift || true
WHAT=${*:-tests ubench stages bench prof pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
fi
if has ubench; then
  timeout 300 tools/ubench_wave > gpurun_out/profiles/${TAG}_ubench_wave.txt 2>&1; echo "ubench rc=$?"
fi
if has stages; then
  timeout 300 python tools/rows_stages.py 4096 > gpurun_out/profiles/${TAG}_rows_stage_ms.txt 2>&1; echo "stages rc=$?"
  cat gpurun_out/profiles/${TAG}_rows_stage_ms.txt
fi
if has bench; then
  timeout 900 python bench.py > gpurun_out/profiles/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
  cat gpurun_out/profiles/${TAG}_bench_line.json
fi
if has prof; then
  bash tools/profile.sh ${TAG}_n4096 4096 > gpurun_out/${TAG}_profile.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/${TAG}_profile.log
fi
if has pmc; then
  bash tools/pmc_wave.sh 4096 ${TAG}_n4096 > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
fi
