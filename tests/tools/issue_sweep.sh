#!/bin/bash
# Throughput of the shipped issue mode (bench.py: 5 slots x 4 requests) under tile / split-K overrides of the backbone convolutions.
# The measured tile table minimises ONE launch's duration on an empty chip; with five slots in flight a tile that fills fewer CUs at a
# higher matrix-pipe efficiency may serve the chip better.   bash tests/tools/issue_sweep.sh <tag>   -> gpurun_out/<tag>_issue_sweep.txt
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
run() {  # name, override
  DD3D_TILE_OVERRIDE="$2" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeat-blocks 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('%-34s value %8.1f  median %8.1f  one-slot %7.1f img/s  bs1 %6.1f img/s  tower %6.1f us' % ('$1', d['value'], d['blocks']['median_images_per_s'], c['images_per_s_one_slot_at_a_time'], c['bs1_images_per_s'], d['roofline']['avg_launch_us']))
" | tee -a $O/${TAG}_issue_sweep.txt
}
L3="30720,128,1152,1"; L4="7680,256,2304,1"; L5="1920,512,4608,1"; L2="122880,64,576,1"; FO="1920+7680+30720,256,2304,1"
run baseline ""
run L3=256x128 "$L3=256x128:1"
run L3=128x128w4 "$L3=128x128w4:1"
run L3=256x128t42 "$L3=256x128t42:1"
run L4=128x128:1 "$L4=128x128:1"
run L4=128x128:2 "$L4=128x128:2"
run L4=128x128w4:1 "$L4=128x128w4:1"
run L5=128x128:2 "$L5=128x128:2"
run L5=128x128:4 "$L5=128x128:4"
run L2=128x64w4 "$L2=128x64w4:1"
run FO=256x128 "$FO=256x128:1"
run big-tiles "$L3=256x128:1;$L4=128x128:1;$L5=128x128:2"
run baseline-again ""
