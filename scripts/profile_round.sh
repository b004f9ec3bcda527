#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench.py command (`python3 bench.py`, the line the driver records);
#   2. PMC passes of the same command, one counter group per run (never combined with tracing
#      beyond --kernel-trace): HBM-side traffic (FETCH_SIZE, WRITE_SIZE), L2 hit rate, SQ busy/
#      wait/VALU counters, L1 counters;
#   3. text summaries via scripts/rocpd_summary.py and profiles/traffic.json for bench.py.
# Usage: bash scripts/profile_round.sh <tag> ["extra bench.py flags" [traffic.json key]]
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EXTRA=${2:-}
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --strong-spp 0 $EXTRA"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o bench -- python3 bench.py $EXTRA > $OUT/bench_stats.log 2>&1 || { echo "kernel trace failed"; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc group $i ($grp) failed" >> $OUT/failed.txt
done
python3 scripts/rocpd_summary.py $OUT > $OUT/summary.txt
grep -h '^{' $OUT/bench_stats.log > $OUT/bench_line_under_profiler.json
grep -h '^{' $OUT/pmc1.log > $OUT/bench_line_under_pmc.json
# traffic.json entry of this workload (third argument: its key, e.g. level10_3840x2160), merged into the tracked file's copy
KEY=${3:-level8_1920x1080}
python3 scripts/make_traffic_json.py $OUT $KEY profiles/traffic.json > $OUT/traffic.json
# the databases are large (gpurun brings back at most 64 MiB): the summaries above are what is kept
[ "${KEEP_DB:-0}" = "1" ] || find $OUT -name "*.db" -delete
echo done > $OUT/finished.txt
