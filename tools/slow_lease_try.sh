#!/bin/bash
# tools/slow_lease_try.sh TAG — is this lease of the slow kind (round 5's 104 KB lane / 96 KB group kernels run
# 15–20 % slower on one lease in four)?  One process of round 5's build at N = 16 384; if its kernel takes more than 0.77 ms, run the
# A/B the round-5 review asked for (round 5's build against the current one at 16 384 / 32 768 / 65 536, three rounds) and keep it.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/profiles
ms=$(IBFT_MIN_ABI=3 IBFT_GPU_LIB=$ROOT/ab/libibftgpu_r05.so timeout 120 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import go_ibft_amd.verifier as V, go_ibft_amd.simulate as SIM
n = 16384
bv = V.BatchVerifier(max_rows=n)
r = SIM.make_round(bv, n, 600 + n)
bv.set_validators(1, r.addrs, r.power); bv.seals_stage(r.hash32, r.seal65, r.signer20, None)
for _ in range(150): bv.seals_run()
bv.set_kernel_timing(1); bv.last_kernel_ms()
for _ in range(40): bv.seals_run()
ms, k = bv.last_kernel_ms()
print(round(ms / k, 4))
PY
)
echo "lease probe: round-5 build, N = 16 384 cold kernel $ms ms ($(hostname))"
if python3 -c "import sys; sys.exit(0 if float('$ms') > 0.77 else 1)"; then
  echo "SLOW KIND: running the A/B"
  timeout 900 python tools/kernel_ab.py ab/libibftgpu_r05.so go-ibft_amd/csrc/libibftgpu.so 3 16384 32768 65536 w16384 > gpurun_out/profiles/${TAG}_slow_lease_kernel_ab.txt 2>&1
  cat gpurun_out/profiles/${TAG}_slow_lease_kernel_ab.txt
  timeout 120 tools/scatter_probe 2>/dev/null | grep -E "loop body +(16|48|64|96) KB +wavefronts +1024" > gpurun_out/profiles/${TAG}_slow_lease_probe.txt; cat gpurun_out/profiles/${TAG}_slow_lease_probe.txt
fi
