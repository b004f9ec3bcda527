#!/bin/bash
# End-to-end wall time of bin/cli_render on the configs[2] stand-in scene: STL from disk,
# vertex normals, tree, upload, 1000 spp at 1920x1080, PNG.  Usage (GPU box, repo root):
#   bash scripts/cli_end_to_end.sh [samples]
set -e
S=${1:-1000}
mkdir -p gpurun_out/e2e
STL=/tmp/standin8_$$.stl
PT_NO_TORCH=1 python3 -c "
import path_tracing_amd as pta
pta.meshgen.write_stl('$STL', pta.meshgen.standin_scene(8))
"
T0=$(date +%s.%N)
PT_CLI_TIMING=1 ./path_tracing_amd/bin/cli_render --stl $STL --output gpurun_out/e2e/out.png \
    --samples $S --width 1920 --height 1080 --dof-aperture 0.1 --dof-distance 4.5 --gpus 1 > gpurun_out/e2e/cli.log 2>&1
T1=$(date +%s.%N)
echo "started at $T0, process gone at $T1" | tee gpurun_out/e2e/wall_abs.txt
python3 -c "print(\"cli_render $S spp 1920x1080, 1343488 triangles: %.2f s wall\" % ($T1 - $T0))" | tee gpurun_out/e2e/wall.txt
tr "\r" "\n" < gpurun_out/e2e/cli.log | grep -a -o "\[timing\].*" | tee -a gpurun_out/e2e/wall.txt
tail -c 300 gpurun_out/e2e/cli.log
rm -f $STL
