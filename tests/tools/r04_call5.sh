#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
for v in "" abl_EPI abl_EPI_STORE abl_MFMA; do
  echo "== ${v:-shipped}" | tee -a $O/epilogue_ablation_b4.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 towers.0 fpn_output3 level3.tree1.tree1.conv2 level2.tree2.conv2 2>&1 | grep " us " | tee -a $O/epilogue_ablation_b4.txt
done
bash tests/tools/r04_tower_pmc.sh 4 gpurun_out/r04f/r04_tower_f16x2_pmc.txt; head -60 $O/r04_tower_f16x2_pmc.txt
