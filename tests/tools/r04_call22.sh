#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04v
mkdir -p $O
cd $R
export DD3D_AMAX=0 DD3D_EXPLORE_MAXBLOCKS=100000
( time DD3D_EXP=dd3d_kitti_v99 timeout 900 python tests/gpu_tile_explore.py 384 1280 16 > $O/tile_explore_v99_b16_all.txt 2>&1 ) 2>&1 | grep real
( time DD3D_EXP=dd3d_nusc_v99 timeout 900 python tests/gpu_tile_explore.py 896 1600 6 > $O/tile_explore_nusc_v99_b6_all.txt 2>&1 ) 2>&1 | grep real
( time DD3D_EXP=dd3d_kitti_v99 timeout 900 python tests/gpu_tile_explore.py 384 1280 4 > $O/tile_explore_v99_b4_all.txt 2>&1 ) 2>&1 | grep real
cp gpurun_out/tile_table_dd3d_*v99*.json $O/
timeout 600 python tests/gpu_configs_check.py v99 2>&1 | grep dd3d_ | cut -c1-200 | tee $O/configs_before_merge.txt
