#!/bin/bash
# Dev helper (GPU box): one bench line per workload (short runs), summarised.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in metal_glass_plates dragon_metal marching_dragons camera_around_david grass_desert; do
  timeout -k 10 500 python -u bench.py --workload $w --steps ${STEPS:-5} --warmup 1 "$@" 2>gpurun_out/bench_$w.err > gpurun_out/bench_$w.json
  python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/bench_{w}.json").read().strip().splitlines()[-1])
    c = j.get("cpu_baseline") or {}
    if "roofline" not in j:          # the sequence workload: frames of an animation, no single frame's roofline
        print(f"{w:22s} {j['config'].get('resolution')} frames {j['config'].get('frames')} ms/frame {j['ms_per_step']:.2f} "
              f"Mrays/s {j['value']:.1f} cpu_ms {c.get('ms_per_frame')}")
        sys.exit(0)
    r = j["roofline"]
    print(f"{w:22s} {j['config']['resolution']} rays/frame {j['config']['rays_per_step']:.3g} ms/frame {j['ms_per_step']:.2f} "
          f"Mrays/s {j['value']:.1f} frac {r['frac']:.3f} agree {r['counters_agree']} cpu_ms {c.get('ms_per_frame')} "
          f"speedup {j['config'].get('speedup_vs_cpu_reference_ms_per_frame')}")
except Exception as e:
    print(w, "FAILED", e, open(f"gpurun_out/bench_{w}.err").read()[-400:])
PY
done
