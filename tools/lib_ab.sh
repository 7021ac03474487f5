#!/bin/bash
# tools/lib_ab.sh OLD.so NEW.so [rounds] — the same bench legs on ONE box, alternating between two builds of libibftgpu.so
# (IBFT_GPU_LIB).  Prints, per run: headline verifies/s, kernel ms, warm kernel ms, and the sweep's cold / warm kernel ms.
OLD=$1; NEW=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for which in old new; do
    lib=$OLD; [ $which = new ] && lib=$NEW
    IBFT_GPU_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-host-mirror --no-certificates --no-sequence --extended-steps 200 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1])
sw=' '.join('%d:%.4f/%.4f'%(e['validators'],e['cold']['kernel_ms'],e['warm']['kernel_ms']) for e in r['sweep']['sizes'])
print('$which', 'value %.3fM'%(r['value']/1e6), 'kernel %.4f'%r['roofline']['avg_kernel_ms'], 'ext %.3fM'%(r['extended']['value']/1e6), 'warm %.4f'%r['warm_path']['kernel_ms'], '|', sw)
"
  done
done
