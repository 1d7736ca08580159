#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_planes_gpu.py tests/test_conv_gpu.py -q -m gpu 2>&1 | tail -5 | tee $O/pytest_conv.txt
( time timeout 900 python tests/gpu_tile_explore.py 384 1280 4 > $O/tile_explore_dla34_b4_f16x2.txt 2>&1 ) 2>&1 | tail -3
cp gpurun_out/tile_table_*b4*.json $O/ 2>/dev/null
tail -3 $O/tile_explore_dla34_b4_f16x2.txt
