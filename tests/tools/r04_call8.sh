#!/bin/bash
# configs at full size, pipeline sweep, tile exploration of the merged FPN launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 900 python tests/gpu_configs_check.py 2>&1 | grep dd3d_ | cut -c1-220 | tee $O/configs.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "5 5 4" "4 4 4" "6 6 4" "5 5 5" "4 4 6" "3 3 8" "4 4 8" "6 6 3" "8 6 2"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$(($2+1)) timeout 200 python $R/bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --repeat-blocks 3 --pipeline $1 --compute-streams $2 --microbatch $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('p$1 cs$2 m$3', d['value'], d['blocks']['ms_per_step'], d['roofline']['frac'], d['config']['bs1_ms_per_image'])" | tee -a $O/pipeline_sweep.txt
done
cd $R
DD3D_AMAX=0 timeout 600 python tests/gpu_tile_explore.py 384 1280 4 > $O/tile_explore_dla34_b4.txt 2>&1; cp gpurun_out/tile_table_*b4*.json $O/; grep "fpn_outputs\|predictors" $O/tile_explore_dla34_b4.txt | cut -c1-300
DD3D_AMAX=0 timeout 600 python tests/gpu_tile_explore.py 384 1280 1 > $O/tile_explore_dla34_b1.txt 2>&1; cp gpurun_out/tile_table_*b1*.json $O/; grep "fpn_outputs\|predictors" $O/tile_explore_dla34_b1.txt | cut -c1-300
