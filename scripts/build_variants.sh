#!/bin/bash
# Builds tuning variants of libpt_hip.so into variants/<name>/libpt_hip.so.
# Usage: bash scripts/build_variants.sh name1="-DFOO=1 -DBAR=2" name2="..."
set -e
cd "$(dirname "$0")/.."
FLAGS="-std=c++17 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC --offload-arch=gfx950 -Iinclude -Ipath_tracing_amd/csrc -Wno-unused-function"
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  mkdir -p variants/$name
  /opt/rocm/bin/hipcc $FLAGS $defs -shared -o variants/$name/libpt_hip.so path_tracing_amd/csrc/pt_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "k_traceILi1" | grep -E "VGPRs:|Scratch|Occupancy" | sed "s/^.*remark: [^ ]* /  $name: /; s/\[-Rpass.*//" &
done
wait
