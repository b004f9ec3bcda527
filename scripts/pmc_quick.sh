#!/bin/bash
# Quick SQ counter passes over the trace micro-benchmark (two runs, one counter group each).
# Usage (GPU box, repo root): bash scripts/pmc_quick.sh <tag> [extra raycast_bench args]
set -u
TAG=${1:-q}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $OUT/g$i -o pmc -- python3 scripts/raycast_bench.py --reps 2 "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/failed.txt
done
python3 scripts/rocpd_summary.py $OUT --kernels k_trace > $OUT/summary.txt
[ "${KEEP_DB:-0}" = "1" ] || find $OUT -name "*.db" -delete
