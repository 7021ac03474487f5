#!/bin/bash
# tools/lib_pipe_ab.sh OLD.so NEW.so [rounds] — the driver's own bench command on ONE lease, alternating between two builds of
# libibftgpu.so (IBFT_GPU_LIB): headline / extended step and kernel, warm path, the sweep's cold and warm M verifies/s.
OLD=$1; NEW=$2; R=${3:-2}
for i in $(seq 1 $R); do for which in old new; do
lib=$OLD; [ $which = new ] && lib=$NEW
IBFT_GPU_LIB=$(readlink -f $lib) timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-certificates --no-host-mirror --no-sequence --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$which', 'value', round(d['value']/1e6,3), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'extended', d.get('extended',{}).get('ms_per_step'), d.get('extended',{}).get('avg_kernel_ms'), 'warm', d.get('warm_path',{}).get('ms_per_step'), d.get('warm_path',{}).get('kernel_ms'), 'sweep', [(e[0], round(e[1]/1e6,3), round(e[3]/1e6,2)) for e in d.get('sweep',[])])"
done; done
