#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/: kernel trace + stats of bench.py, and
# PMC passes of the trace micro-benchmark (each counter group in a run of its own).
# Usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/ray_stats -o ray -- python3 scripts/raycast_bench.py --reps 3 > $OUT/ray_stats.log 2>&1 || exit 1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp -d $OUT/pmc_$name -o pmc -- python3 scripts/raycast_bench.py --reps 2 --sets primary,sphere > $OUT/pmc_$name.log 2>&1 || echo "pmc group $name failed" >> $OUT/failed.txt
done
echo done > $OUT/finished.txt
