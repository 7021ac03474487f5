#!/bin/bash
# tools/numa_ab.sh — does the host side of a box decide what round 4 called "two classes of boxes"?  The same process pinned to
# the CPUs of each NUMA node in turn: the warm message-set sequence (host-latency-bound: gather from pinned host memory, results
# into mapped host memory) and the N = 16 384 / 65 536 cold kernels (3 056 B of scratch per lane).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "# GPU numa_node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
lscpu | grep -i "numa\|model name\|^CPU(s)"
echo "# affinity: $(python3 -c 'import os; a=sorted(os.sched_getaffinity(0)); print(len(a), a[0], a[-1])')   cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
env | grep -i "^HSA_\|^HIP_\|^ROCR\|^GPU_" 
for node in $(lscpu | grep -i "numa node[0-9]* cpu" | sed 's/.*:\s*//' ); do
  echo "## taskset -c $node"
  taskset -c "$node" python3 tools/seq_breakdown.py 200 pinned 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
for p in ('cold','warm'):
    print(p, 'set_prepare', d[p]['set_prepare'], 'set_commit', d[p]['set_commit'])"
  taskset -c "$node" python3 tools/two_waves_ab.py --rounds 2 --steps 20 2>/dev/null | grep -E "^ +[0-9]"
  IBFT_COLD_LANES= taskset -c "$node" python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.pop("IBFT_COLD_LANES", None)
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
for n in (16384, 65536):
    bv = V.BatchVerifier(flags=0, max_rows=n)
    r = SIM.make_round(bv, n, 100 + n)
    bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
    for _ in range(120): bv.seals_run()
    bv.set_kernel_timing(1); bv.last_kernel_ms()
    t0 = time.perf_counter()
    for _ in range(40): bv.seals_run()
    el = time.perf_counter() - t0
    ms, k = bv.last_kernel_ms()
    print(f"cold N={n}: kernel {ms/k:.4f} ms, step {el/40*1e3:.4f} ms, lanes {bv.last_dispatch()[0]}")
    bv.close()
PY
done
