#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parallel_gpu.py -q -m gpu -x -k "rccl or bench" 2>&1 | tail -5 | tee $O/pytest_parallel.txt
for v in "" abl_DMA abl_MFMA; do
  echo "== ${v:-shipped}" | tee -a $O/ablation_real_operands_b4.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 level2.tree2.conv2 level3.tree1.tree1.conv2 level4.tree1.tree1.conv2 level5.tree1.conv2 fpn_outputs towers.0 level3.tree2.root 2>&1 | grep " us " | tee -a $O/ablation_real_operands_b4.txt
done
