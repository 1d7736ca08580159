#!/bin/bash
# What bounds the K loop of the backbone-sized convolutions: the shipped kernels against builds that drop one ingredient (wrong results;
# build them with: for v in DMA MFMA DSREAD EPI; do bash tests/tools/build_variant.sh abl_$v --ablations -DDD3D_ABLATE_$v; done).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
export DD3D_AMAX=0
for v in "" abl_DMA abl_MFMA abl_DSREAD abl_EPI; do
  echo "== ${v:-shipped}" | tee -a $O/ablation_b4.txt
  DD3D_HIP_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 2>&1 | grep " us " | tee -a $O/ablation_b4.txt
done
timeout 600 python tests/gpu_tile_explore.py 384 1280 4 > $O/tile_explore_dla34_b4_f16x2.txt 2>&1; cp gpurun_out/tile_table_*b4*.json $O/; tail -2 $O/tile_explore_dla34_b4_f16x2.txt
timeout 600 python tests/gpu_tile_explore.py 384 1280 1 > $O/tile_explore_dla34_b1_f16x2.txt 2>&1; cp gpurun_out/tile_table_*b1*.json $O/; tail -2 $O/tile_explore_dla34_b1_f16x2.txt
