#!/bin/bash
# tools/gpu_session_r04.sh PART... — round-4 GPU sessions (one gpurun call each); everything lands under gpurun_out/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/profiles
has() { [[ " $* " == *" $PART "* ]]; }
for PART in "$@"; do
  case $PART in
    newtests)
      timeout 1500 python -m pytest tests/test_gpu_prepare_quorum.py tests/test_gpu_comm.py tests/test_gpu_host.py tests/test_gpu_message_set.py tests/test_gpu_tally_wide.py -m gpu -x -q > gpurun_out/r04a_pytest_new.log 2>&1
      echo "newtests rc=$?"; tail -5 gpurun_out/r04a_pytest_new.log ;;
    occupancy)
      timeout 900 python tools/occupancy.py --steps 30 > gpurun_out/profiles/r04_occupancy.txt 2> gpurun_out/r04_occupancy.err
      echo "occupancy rc=$?"; cat gpurun_out/profiles/r04_occupancy.txt; tail -3 gpurun_out/r04_occupancy.err ;;
    profsizes)
      timeout 1500 bash tools/profile_sizes.sh r04 ${IBFT_PROF_SIZES:-512 1024 16384 65536} > gpurun_out/r04_profile_sizes.log 2>&1
      echo "profsizes rc=$?"; tail -40 gpurun_out/r04_profile_sizes.log ;;
    config5)
      IBFT_BENCH_FORCE_DIST=1 IBFT_BENCH_CONFIG5=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 timeout 600 python bench.py --steps 20 --warmup 5 \
        --no-cpu-baseline --no-sequence --no-warm --no-sweep --no-sustained --no-certificates --no-host-mirror --extended-steps 0 \
        > gpurun_out/profiles/r04_forcedist_config5.json 2> gpurun_out/r04_forcedist_config5.err
      echo "config5 rc=$?"; tail -c 1500 gpurun_out/profiles/r04_forcedist_config5.json; tail -3 gpurun_out/r04_forcedist_config5.err ;;
    pair)
      timeout 600 python tools/pair_ab.py > gpurun_out/profiles/r04_pair_ab.txt 2> gpurun_out/r04_pair_ab.err
      echo "pair rc=$?"; cat gpurun_out/profiles/r04_pair_ab.txt; tail -3 gpurun_out/r04_pair_ab.err
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_message_set.py tests/test_gpu_arith.py -m gpu -x -q > gpurun_out/r04_pytest_pair.log 2>&1
      echo "pairtests rc=$?"; tail -4 gpurun_out/r04_pytest_pair.log ;;
    soak)
      timeout 1200 python tools/soak.py > gpurun_out/profiles/r04_soak.json 2> gpurun_out/r04_soak.err
      echo "soak rc=$?"; tail -c 600 gpurun_out/profiles/r04_soak.json; tail -3 gpurun_out/r04_soak.err ;;
    alltests)
      timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest_gpu.log 2>&1
      echo "alltests rc=$?"; tail -5 gpurun_out/r04_pytest_gpu.log ;;
    bench)
      timeout 900 python bench.py > gpurun_out/profiles/r04_bench_line.json 2> gpurun_out/r04_bench.err
      echo "bench rc=$?"; tail -c 3000 gpurun_out/profiles/r04_bench_line.json ;;
    prof4096)
      bash tools/profile.sh r04_n4096 4096 > gpurun_out/r04_profile.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/r04_profile.log
      bash tools/pmc_wave.sh 4096 r04_n4096 > gpurun_out/r04_pmc.log 2>&1; echo "pmc rc=$?" ;;
    *) echo "unknown part $PART" ;;
  esac
done
