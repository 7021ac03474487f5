#!/bin/bash
# tools/gpu_session_r06.sh TAG [what...] — one gpurun call of round 6.
#   tests     the GPU suite (the full 10 312-round soak is part of it now)
#   drv       the driver's own bench command (python3 bench.py --gpus 1 --steps 20 --warmup 5): stdout kept, last line parsed;
#             the counter files its own rocprofv3 sub-steps wrote are kept next to it
#   ab        tools/kernel_ab.py ab/libibftgpu_r05.so <current> (same lease, alternating processes)
#   forcedist IBFT_BENCH_FORCE_DIST=1 bench (one-rank RCCL communicator + the sharded sweep as one shard)
#   smalln    tools/small_n.py (N = 4, 6, 30 … cold / warm next to one CPU core: the crossover of the min-device-rows knob)
#   profile   python bench.py --profile (the full counter series)
#   stages    tools/rows_stages.py
#   sizes     tools/profile_sizes.sh for 16384 65536 (stats + traffic + counters)
#   soak      tools/soak.py
#   smoke     __graft_entry__.smoke()
set -u
TAG=${1:-r06a}
shift || true
WHAT=${*:-tests drv ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/profiles/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/profiles/${TAG}_smoke.log
fi
if has tests; then
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/profiles/${TAG}_pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/profiles/${TAG}_pytest_gpu.log
  [ -e gpurun_out/soak_in_suite.json ] && cp gpurun_out/soak_in_suite.json gpurun_out/profiles/${TAG}_soak_in_suite.json
fi
if has drv; then
  t0=$(date +%s)
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_stdout.txt 2> gpurun_out/${TAG}_driver_stderr.txt
  echo "driver-style bench rc=$? in $(( $(date +%s) - t0 )) s"
  tail -c 8081 gpurun_out/${TAG}_driver_stdout.txt > gpurun_out/${TAG}_driver_tail8081.txt     # what the driver keeps
  python3 - gpurun_out/${TAG}_driver_tail8081.txt gpurun_out/profiles/${TAG}_bench_line.json <<'PY'
import json, sys
tail = open(sys.argv[1]).read()
last = [l for l in tail.splitlines() if l.strip()][-1]
rec = json.loads(last)
open(sys.argv[2], "w").write(last + "\n")
rf = rec["roofline"]
print(f"last stdout line: {len(last)} bytes, parses; value {rec['value']:.0f} {rec['unit']}, ms_per_step {rec['ms_per_step']:.4f}, "
      f"kernel {rf['avg_kernel_ms']:.4f} ms (rocprof {rf.get('rocprof_avg_kernel_ms')}), traffic {rf.get('traffic')}, counters_live {rf.get('counters_live')}, "
      f"canary {rec.get('device_canary')}, extended {rec.get('extended')}, quorum p50 {rec.get('quorum_latency_ms_p50')}, cpu {rec.get('cpu_baseline', {}).get('value')}")
print("counters:", rf.get("counters"))
PY
  cp gpurun_out/bench_detail.json gpurun_out/profiles/${TAG}_bench_detail.json 2>/dev/null
  for f in gpurun_out/profiles/live_n4096_*; do [ -e "$f" ] && cp "$f" "gpurun_out/profiles/${TAG}_drv_$(basename $f)"; done
  tail -5 gpurun_out/${TAG}_driver_stderr.txt
fi
if has ab; then
  timeout 1500 python tools/kernel_ab.py ${IBFT_AB_OLD:-ab/libibftgpu_r05.so} ${IBFT_AB_NEW:-go-ibft_amd/csrc/libibftgpu.so} ${IBFT_AB_ROUNDS:-3} ${IBFT_AB_SIZES:-} > gpurun_out/profiles/${TAG}_kernel_ab.txt 2> gpurun_out/${TAG}_kernel_ab.err
  echo "kernel A/B rc=$?"; cat gpurun_out/profiles/${TAG}_kernel_ab.txt; tail -3 gpurun_out/${TAG}_kernel_ab.err
fi
if has forcedist; then
  IBFT_BENCH_FORCE_DIST=1 IBFT_BENCH_SHARDED_SWEEP=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_forcedist_stdout.txt 2> gpurun_out/${TAG}_forcedist_stderr.txt
  echo "forced-dist bench rc=$?"; tail -1 gpurun_out/${TAG}_forcedist_stdout.txt > gpurun_out/profiles/${TAG}_forcedist_line.json
  cat gpurun_out/profiles/${TAG}_forcedist_line.json; tail -3 gpurun_out/${TAG}_forcedist_stderr.txt
fi
if has smalln; then
  timeout 900 python tools/small_n.py > gpurun_out/profiles/${TAG}_small_n.json 2> gpurun_out/${TAG}_small_n.err; echo "small-N rc=$?"
  cat gpurun_out/profiles/${TAG}_small_n.json; tail -3 gpurun_out/${TAG}_small_n.err
fi
if has profile; then
  timeout 1500 python bench.py --profile --no-sweep --no-certificates --no-host-mirror > gpurun_out/${TAG}_profile_stdout.txt 2> gpurun_out/${TAG}_profile_stderr.txt
  echo "bench --profile rc=$?"
  tail -1 gpurun_out/${TAG}_profile_stdout.txt > gpurun_out/profiles/${TAG}_profile_line.json
  cat gpurun_out/profiles/${TAG}_profile_line.json; tail -5 gpurun_out/${TAG}_profile_stderr.txt
  for f in gpurun_out/profiles/live_n4096_*; do [ -e "$f" ] && cp "$f" "gpurun_out/profiles/${TAG}_profile_$(basename $f)"; done
fi
if has stages; then
  timeout 300 python tools/rows_stages.py 4096 > gpurun_out/profiles/${TAG}_rows_stage_ms.txt 2>&1; echo "stages rc=$?"; cat gpurun_out/profiles/${TAG}_rows_stage_ms.txt
fi
if has sizes; then
  timeout 1500 bash tools/profile_sizes.sh $TAG ${IBFT_PROF_SIZES:-16384 65536} > gpurun_out/${TAG}_profile_sizes.log 2>&1
  echo "profsizes rc=$?"; tail -5 gpurun_out/${TAG}_profile_sizes.log
fi
if has soak; then
  timeout 1500 python tools/soak.py > gpurun_out/profiles/${TAG}_soak.json 2> gpurun_out/${TAG}_soak.err
  echo "soak rc=$?"; tail -c 500 gpurun_out/profiles/${TAG}_soak.json; tail -2 gpurun_out/${TAG}_soak.err
fi
if has ubench; then
  timeout 600 tools/ubench_wave > gpurun_out/profiles/${TAG}_ubench_wave.txt 2>&1; echo "ubench rc=$?"
  grep -E "^class" gpurun_out/profiles/${TAG}_ubench_wave.txt | head -8
fi
if has parity; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_warm_path.py tests/test_gpu_arith.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/profiles/${TAG}_pytest_parity.log 2>&1
  echo "parity tests rc=$?"; tail -3 gpurun_out/profiles/${TAG}_pytest_parity.log
fi
if has fresh; then
  timeout 1500 python tools/fresh_batch_ab.py ${IBFT_AB_OLD:-ab/libibftgpu_r05.so} ${IBFT_AB_NEW:-go-ibft_amd/csrc/libibftgpu.so} ${IBFT_AB_ROUNDS:-3} > gpurun_out/profiles/${TAG}_fresh_batch_ab.txt 2> gpurun_out/${TAG}_fresh_ab.err
  echo "fresh-batch A/B rc=$?"; cat gpurun_out/profiles/${TAG}_fresh_batch_ab.txt; tail -3 gpurun_out/${TAG}_fresh_ab.err
fi
if has stageissue; then
  timeout 600 python tools/rows_stage_issue.py 4096 > gpurun_out/profiles/${TAG}_rows_stage_issue.txt 2> gpurun_out/${TAG}_stage_issue.err; echo "stage issue rc=$?"
  cat gpurun_out/profiles/${TAG}_rows_stage_issue.txt; tail -3 gpurun_out/${TAG}_stage_issue.err
fi
if has gpre; then
  # the row recover's stage times with the G-table prefetch (IBFT_ROWS_G_PREFETCH=1: ab/libibft_devtest_gpre.so) and without, alternating
  for r in 1 2 3; do
    for which in off on; do
      so=""; [ $which = on ] && so="$ROOT/ab/libibft_devtest_gpre.so"
      echo "== prefetch $which (round $r)"; DEVTEST_SO=$so timeout 300 python tools/rows_stages.py 4096 2>&1 | grep -E "main loop|G additions|Z\^-1|complete|tables"
    done
  done > gpurun_out/profiles/${TAG}_gpre_stage_ab.txt 2>&1
  echo "gpre rc=$?"; cat gpurun_out/profiles/${TAG}_gpre_stage_ab.txt
fi
