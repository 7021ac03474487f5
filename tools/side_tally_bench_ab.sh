#!/bin/bash
# tools/side_tally_bench_ab.sh — the driver's own bench command with the tally on one stream (IBFT_SIDE_TALLY=0) and with the
# library's AUTO rule (unset), alternating, one lease: headline, extended leg, warm path and the sweep's cold / warm M verifies/s
for r in 1 2; do for s in 0 auto; do
if [ $s = auto ]; then unset IBFT_SIDE_TALLY; else export IBFT_SIDE_TALLY=$s; fi
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-certificates --no-host-mirror --no-sequence --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('side=$s', 'value', round(d['value']/1e6,3), round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'extended', d.get('extended',{}).get('ms_per_step'), d.get('extended',{}).get('avg_kernel_ms'), 'warm', d.get('warm_path',{}).get('ms_per_step'), d.get('warm_path',{}).get('kernel_ms'), 'sweep', [(e[0], round(e[1]/1e6,3), round(e[3]/1e6,2)) for e in d.get('sweep',[])])"
done; done
