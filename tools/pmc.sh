#!/bin/bash
# Dev helper (GPU box): hardware-counter passes over a short bench run, one rocprofv3
# invocation per counter group.  Usage: tools/pmc.sh <tag> [bench args]
cd "$(dirname "$0")/.."
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE TCC_HIT_sum" \
           "TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out -o p$i -- python3 $OLDPWD/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
ls $out
