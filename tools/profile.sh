#!/bin/bash
# Dev helper (GPU box): the round's profile of one workload -- a bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the hardware-counter passes -- into gpurun_out/prof_<tag>/.
# Usage: tools/profile.sh <tag> <workload> [more bench args]
set -o pipefail
cd "$(dirname "$0")/.."
repo=$PWD
tag=$1; wl=$2; shift 2
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $repo/bench.py --workload $wl --steps 20 --warmup 5 --no-extras "$@" > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
    python3 $repo/bench.py --workload $wl --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
cp $(find $out/stats -name "*kernel_stats.csv" | head -n 1) $out/kernel_stats.csv
$repo/tools/pmc.sh $tag --workload $wl --no-extras "$@" > $out/pmc.log 2>&1
python3 $repo/tools/pmc_to_json.py $repo/gpurun_out/pmc_$tag $wl $out/pmc.json "${RTX_BUILD_ID:-?}" > $out/pmc_summary.txt 2>&1
cat $out/kernel_stats.csv | head -n 12
