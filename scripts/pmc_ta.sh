#!/bin/bash
# Vector-memory-path counters of the default bench (texture addresser / L1 busy and stall cycles),
# one counter group per run.  Usage (GPU box, repo root): bash scripts/pmc_ta.sh <tag> ["extra bench.py flags"]
set -u
TAG=${1:-ta}
EXTRA=${2:-}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp -d $OUT/g$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --strong-spp 0 $EXTRA > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed" >> $OUT/failed.txt
done
python3 scripts/rocpd_summary.py $OUT --kernels k_trace > $OUT/summary.txt
[ "${KEEP_DB:-0}" = "1" ] || find $OUT -name "*.db" -delete
