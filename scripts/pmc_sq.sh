#!/bin/bash
# SQ instruction-mix counters of the default bench, one group per run.
# Usage (GPU box, repo root): bash scripts/pmc_sq.sh <tag> ["extra bench.py flags"]
set -u
TAG=${1:-sq}
EXTRA=${2:-}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp -d $OUT/g$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --strong-spp 0 $EXTRA > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed" >> $OUT/failed.txt
done
python3 scripts/rocpd_summary.py $OUT --kernels k_trace,k_shadow_dense,k_shadow_leafwise > $OUT/summary.txt
[ "${KEEP_DB:-0}" = "1" ] || find $OUT -name "*.db" -delete
