#!/bin/bash
# tools/gpu_session_r05.sh TAG [what...] — one gpurun call of round 5.
#   tests    the GPU suite
#   drv      the driver's own bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5): stdout kept, last line parsed
#   ubench   tools/ubench_wave (issue times per instruction class and alignment, 1 / 2 / 4 wavefronts per SIMD)
#   twowaves tools/two_waves_ab.py
#   profile  python bench.py --profile (rocprofv3 sub-steps of the same run; the line and its counter files kept)
#   forcedist IBFT_BENCH_FORCE_DIST=1 IBFT_BENCH_CONFIG5=1 bench (one-rank RCCL communicator)
#   newtests  the GPU tests added this round
#   tables   tools/table_ab.py (LDS / private-segment window tables)
#   soak     tools/soak.py (10 300 Byzantine rounds against the oracle)
#   stages   tools/rows_stages.py
#   sizes    tools/profile_sizes.sh for 16384 65536 (stats + traffic + counters)
set -u
TAG=${1:-r05a}
shift || true
WHAT=${*:-tests drv ubench twowaves profile}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/profiles/${TAG}_pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/profiles/${TAG}_pytest_gpu.log
fi
if has drv; then
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_stdout.txt 2> gpurun_out/${TAG}_driver_stderr.txt
  echo "driver-style bench rc=$?"
  tail -c 8081 gpurun_out/${TAG}_driver_stdout.txt > gpurun_out/${TAG}_driver_tail8081.txt     # what the driver keeps
  python3 - gpurun_out/${TAG}_driver_tail8081.txt gpurun_out/profiles/${TAG}_bench_line.json <<'PY'
import json, sys
tail = open(sys.argv[1]).read()
last = [l for l in tail.splitlines() if l.strip()][-1]
rec = json.loads(last)
open(sys.argv[2], "w").write(last + "\n")
print(f"last stdout line: {len(last)} bytes, parses; value {rec['value']:.0f} {rec['unit']}, ms_per_step {rec['ms_per_step']:.4f}, "
      f"kernel {rec['roofline']['avg_kernel_ms']:.4f} ms, quorum p50 {rec.get('quorum_latency_ms_p50')}, cpu {rec.get('cpu_baseline', {}).get('value')}")
PY
  cp gpurun_out/bench_detail.json gpurun_out/profiles/${TAG}_bench_detail.json 2>/dev/null
fi
if has ubench; then
  timeout 600 tools/ubench_wave > gpurun_out/profiles/${TAG}_ubench_wave.txt 2>&1; echo "ubench rc=$?"
  grep -E "^class|wfe_mul<inline>, 4" gpurun_out/profiles/${TAG}_ubench_wave.txt
fi
if has twowaves; then
  timeout 600 python tools/two_waves_ab.py > gpurun_out/profiles/${TAG}_two_waves_ab.txt 2> gpurun_out/${TAG}_two_waves.err; echo "twowaves rc=$?"
  cat gpurun_out/profiles/${TAG}_two_waves_ab.txt; tail -3 gpurun_out/${TAG}_two_waves.err
fi
if has profile; then
  timeout 1500 python bench.py --profile --no-sweep --no-certificates --no-host-mirror > gpurun_out/${TAG}_profile_stdout.txt 2> gpurun_out/${TAG}_profile_stderr.txt
  echo "bench --profile rc=$?"
  tail -1 gpurun_out/${TAG}_profile_stdout.txt > gpurun_out/profiles/${TAG}_profile_line.json
  cat gpurun_out/profiles/${TAG}_profile_line.json; tail -5 gpurun_out/${TAG}_profile_stderr.txt
  for f in gpurun_out/profiles/live_n4096_*; do [ -e "$f" ] && cp "$f" "gpurun_out/profiles/${TAG}_profile_$(basename $f)"; done
fi
if has forcedist; then
  # the sharded code path on one GPU: gloo group of one rank + a real one-rank RCCL communicator (the image's librccl), then config #5's shape
  IBFT_BENCH_FORCE_DIST=1 IBFT_BENCH_CONFIG5=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_forcedist_stdout.txt 2> gpurun_out/${TAG}_forcedist_stderr.txt
  echo "forced-dist bench rc=$?"; tail -1 gpurun_out/${TAG}_forcedist_stdout.txt > gpurun_out/profiles/${TAG}_forcedist_config5.json
  cat gpurun_out/profiles/${TAG}_forcedist_config5.json; tail -3 gpurun_out/${TAG}_forcedist_stderr.txt
fi
if has newtests; then
  timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_prepare_quorum.py tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/profiles/${TAG}_pytest_new.log 2>&1
  echo "new tests rc=$?"; tail -3 gpurun_out/profiles/${TAG}_pytest_new.log
fi
if has tables; then
  timeout 600 python tools/table_ab.py > gpurun_out/profiles/${TAG}_table_ab.txt 2> gpurun_out/${TAG}_table_ab.err; echo "table A/B rc=$?"
  cat gpurun_out/profiles/${TAG}_table_ab.txt
fi
if has soak; then
  timeout 1500 python tools/soak.py > gpurun_out/profiles/${TAG}_soak.json 2> gpurun_out/${TAG}_soak.err
  echo "soak rc=$?"; tail -c 500 gpurun_out/profiles/${TAG}_soak.json; tail -2 gpurun_out/${TAG}_soak.err
fi
if has stages; then
  timeout 300 python tools/rows_stages.py 4096 > gpurun_out/profiles/${TAG}_rows_stage_ms.txt 2>&1; echo "stages rc=$?"
fi
if has sizes; then
  timeout 1500 bash tools/profile_sizes.sh $TAG ${IBFT_PROF_SIZES:-16384 65536} > gpurun_out/${TAG}_profile_sizes.log 2>&1
  echo "profsizes rc=$?"; tail -5 gpurun_out/${TAG}_profile_sizes.log
fi
