#!/bin/bash
# tools/lib_ab_many.sh ROUNDS LIB... — the headline leg only (N = 4 096 cold, 200 + 200 steps) on ONE box, the builds of
# libibftgpu.so taken in turn, ROUNDS times over: value, verdict-kernel ms (HIP events)
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do
    IBFT_GPU_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-host-mirror --no-certificates --no-sequence --no-sweep --no-warm --extended-steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$lib', 'value %.3fM'%(r['value']/1e6), 'kernel %.4f'%r['roofline']['avg_kernel_ms'], 'ext %.3fM'%(r['extended']['value']/1e6), 'ext kernel %.4f'%r['extended']['avg_kernel_ms'])
"
  done
done
