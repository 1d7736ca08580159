#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04t
mkdir -p $O
cd $R
export DD3D_AMAX=0 DD3D_EXPLORE_ONLY=towers.0,fpn_outputs,stem_2,OSA2_1 DD3D_EXPLORE_MAXBLOCKS=100000
DD3D_EXP=dd3d_kitti_v99 timeout 900 python tests/gpu_tile_explore.py 384 1280 16 2>&1 | grep -v "build\|amdgpu" | cut -c1-400 | tee $O/tile_explore_v99_b16_big.txt
DD3D_EXP=dd3d_nusc_v99 timeout 900 python tests/gpu_tile_explore.py 896 1600 6 2>&1 | grep -v "build\|amdgpu" | cut -c1-400 | tee $O/tile_explore_nusc_v99_b6_big.txt
DD3D_EXP=dd3d_nusc_dla34 timeout 900 python tests/gpu_tile_explore.py 896 1600 6 2>&1 | grep -v "build\|amdgpu" | cut -c1-400 | tee $O/tile_explore_nusc_dla34_b6_big.txt
DD3D_EXPLORE_ONLY=towers.0 DD3D_EXP=dd3d_kitti_dla34 timeout 900 python tests/gpu_tile_explore.py 384 1280 8 2>&1 | grep -v "build\|amdgpu" | cut -c1-400 | tee $O/tile_explore_dla34_b8_big.txt
cp gpurun_out/tile_table_*_only.json $O/
