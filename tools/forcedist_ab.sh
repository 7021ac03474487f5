#!/bin/bash
# tools/forcedist_ab.sh OLD.so NEW.so [rounds] — a RANK's step (IBFT_BENCH_FORCE_DIST=1: one-rank RCCL communicator, tally + exchange
# in every step) with two builds of the library alternating on one lease: headline ms/step, 400-step ms/step, synchronous step.
OLD=$1; NEW=$2; R=${3:-3}
for i in $(seq 1 $R); do for which in old new; do
lib=$OLD; [ $which = new ] && lib=$NEW
IBFT_MIN_ABI=3 IBFT_BENCH_FORCE_DIST=1 IBFT_GPU_LIB=$(readlink -f $lib) timeout 300 python bench.py --steps 20 --warmup 5 --no-live-counters --no-certificates --no-host-mirror --no-sequence --no-cpu-baseline --no-sustained --no-sweep --no-warm 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$which', 'value', round(d['value']/1e6,3), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'extended', d.get('extended',{}).get('ms_per_step'), 'sync step', d.get('step_latency_ms_p50'), 'rccl ranks', d.get('rccl_nranks'))"
done; done
