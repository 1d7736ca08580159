#!/bin/bash
# bench.py under other slot / micro-batch / stream counts than the default 5 x 4 on 5 streams (same tree, same box).
#   bash tests/tools/pipeline_sweep.sh <tag>  -> gpurun_out/<tag>_pipeline_sweep.txt
TAG=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
run() {  # slots microbatch streams steps
  GPU_MAX_HW_QUEUES=$(( $3 + 1 )) timeout 300 python bench.py --gpus 1 --steps $4 --warmup 5 --no-cpu-baseline --repeat-blocks 3 --pipeline $1 --microbatch $2 --compute-streams $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('slots %2d x microbatch %2d on %d streams, %3d steps: value %8.1f  median %8.1f  bs1 %6.1f img/s  tower %6.1f us' % ($1, $2, $3, $4, d['value'], d['blocks']['median_images_per_s'], c['bs1_images_per_s'], d['roofline']['avg_launch_us']))
" | tee -a $O/${TAG}_pipeline_sweep.txt
}
run 5 4 5 20
run 5 4 5 40
run 4 5 4 20
run 10 2 5 20
run 10 2 8 20
run 7 3 7 21
run 3 7 3 21
run 2 10 2 20
run 1 20 1 20
run 6 4 6 24
run 8 4 8 32
run 5 8 5 40
run 4 8 4 32
run 5 4 5 20
