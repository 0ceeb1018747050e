#!/bin/bash
# Dev helper (GPU box): time bench.py for the default library and every variant under
# raytracer_amd/lib/variants/*.so, twice each, interleaved.  Usage: tools/exp_variants.sh [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "" raytracer_amd/lib/variants/*.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  name=${lib:-default}
  RTX_AMD_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python -u bench.py --steps 15 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null \
    | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms/frame %.3f'%j['ms_per_step'], 'trace %.3f'%j['roofline']['trace_ms_per_frame'], 'Mrays/s %.1f'%j['value'])" || echo "$name FAILED"
done
done
