R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02r
mkdir -p $O
timeout 900 python tests/gpu_tile_explore.py 384 1280 1 > $O/tile_explore_dla34_f16x2_row.txt 2>&1; tail -2 $O/tile_explore_dla34_f16x2_row.txt
DD3D_EXP=dd3d_kitti_v99 timeout 900 python tests/gpu_tile_explore.py 384 1280 1 > $O/tile_explore_v99_b1_f16x2_row.txt 2>&1; tail -2 $O/tile_explore_v99_b1_f16x2_row.txt
cp gpurun_out/tile_table_*planes.json $O/
