R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02u
mkdir -p $O
DD3D_EXP=dd3d_kitti_v99 timeout 600 python tests/gpu_tile_explore.py 384 1280 16 > $O/tile_explore_v99_b16_f16x2_row.txt 2>&1; tail -1 $O/tile_explore_v99_b16_f16x2_row.txt
DD3D_EXP=dd3d_nusc_dla34 timeout 600 python tests/gpu_tile_explore.py 896 1600 6 > $O/tile_explore_nusc_dla34_b6_f16x2_row.txt 2>&1; tail -1 $O/tile_explore_nusc_dla34_b6_f16x2_row.txt
cp gpurun_out/tile_table_*planes.json $O/
