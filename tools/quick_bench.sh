#!/bin/bash
# Dev helper (GPU box): ms/frame of the headline workloads, nothing else.  Usage: tools/quick_bench.sh [workloads...]
cd "$(dirname "$0")/.."
for w in ${@:-dragon_metal grass_desert}; do
  steps=20; [ $w = grass_desert ] && steps=5
  timeout -k 10 300 python3 -u bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --no-extras --scene-view 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j.get('roofline')
        if r is None:
            q=j['sequence']; print('$w', 'ms/frame %.3f' % q['ms_per_frame_per_gpu'], 'Mrays/s %.1f' % j['value'], 'trace ms %.3f' % q['trace_ms_per_frame'], 'kernels ms %.3f' % q['kernel_ms_per_frame']); continue
        print('$w', 'ms/frame %.3f' % j['ms_per_step'], 'Mrays/s %.1f' % j['value'], 'trace ms %.3f' % r['trace_ms_per_frame'], 'kernels ms %.3f' % r['all_kernels_ms_per_frame'], 'parity', j.get('parity',{}).get('vs_oracle'))
" || exit 1
done
