#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
export DD3D_AMAX=0
run() { # exp H W B
  ( time DD3D_EXP=$1 timeout 600 python tests/gpu_tile_explore.py $2 $3 $4 > $O/tile_explore_$1_b$4.txt 2>&1 ) 2>&1 | grep real
  tail -1 $O/tile_explore_$1_b$4.txt
}
run dd3d_kitti_dla34 384 1280 2
run dd3d_kitti_dla34 384 1280 8
run dd3d_kitti_v99 384 1280 1
run dd3d_kitti_v99 384 1280 4
run dd3d_nusc_v99 896 1600 6
run dd3d_nusc_dla34 896 1600 3
cp gpurun_out/tile_table_*.json $O/
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline > $O/bench_serial.json 2>/dev/null; cut -c1-150 $O/bench_serial.json
