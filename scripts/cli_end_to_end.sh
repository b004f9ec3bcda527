#!/bin/bash
# End-to-end wall time of bin/cli_render on a stand-in scene: STL from disk, vertex normals, tree,
# upload, render, PNG.  Usage (GPU box, repo root):
#   bash scripts/cli_end_to_end.sh [samples [mesh level [width height]]]
# Defaults: 1000 spp of the configs[2] scene (level 8, 1 343 488 triangles) at 1920x1080;
# level 10 (21 495 808 triangles) with 3840 2160 is the shape of configs[4].
set -e
S=${1:-1000}
LEVEL=${2:-8}
W=${3:-1920}
H=${4:-1080}
OUT=gpurun_out/e2e_level${LEVEL}
mkdir -p $OUT
STL=/tmp/standin${LEVEL}_$$.stl
PT_NO_TORCH=1 python3 -c "
import path_tracing_amd as pta
t = pta.meshgen.standin_scene($LEVEL)
pta.meshgen.write_stl('$STL', t)
open('$OUT/triangles.txt', 'w').write(str(len(t)))
"
N=$(cat $OUT/triangles.txt)
T0=$(date +%s.%N)
PT_CLI_TIMING=1 ./path_tracing_amd/bin/cli_render --stl $STL --output $OUT/out.png \
    --samples $S --width $W --height $H --dof-aperture 0.1 --dof-distance 4.5 --gpus 1 > $OUT/cli.log 2>&1
T1=$(date +%s.%N)
echo "started at $T0, process gone at $T1" | tee $OUT/wall_abs.txt
python3 -c "print(\"cli_render $S spp ${W}x${H}, $N triangles: %.2f s wall\" % ($T1 - $T0))" | tee $OUT/wall.txt
tr "\r" "\n" < $OUT/cli.log | grep -a -o "\[timing\].*" | tee -a $OUT/wall.txt
tail -c 300 $OUT/cli.log
rm -f $STL $OUT/out.png
