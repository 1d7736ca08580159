#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
export DD3D_AMAX=0
( time DD3D_EXP=dd3d_kitti_v99 timeout 900 python tests/gpu_tile_explore.py 384 1280 16 > $O/tile_explore_v99_b16.txt 2>&1 ) 2>&1 | grep real; cp gpurun_out/tile_table_dd3d_kitti_v99*b16*.json $O/; tail -1 $O/tile_explore_v99_b16.txt
( time DD3D_EXP=dd3d_nusc_dla34 timeout 600 python tests/gpu_tile_explore.py 896 1600 6 > $O/tile_explore_nusc_dla34_b6.txt 2>&1 ) 2>&1 | grep real; cp gpurun_out/tile_table_dd3d_nusc_dla34*b6*.json $O/; tail -1 $O/tile_explore_nusc_dla34_b6.txt
