#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
for v in "" abl_DSREAD_LOOP; do
  echo "== ${v:-shipped}" | tee -a $O/no_fragment_reads_after_the_first_group_b4.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 level2.tree2.conv2 level3.tree1.tree1.conv2 level4.tree1.tree1.conv2 level5.tree1.conv2 fpn_outputs towers.0 2>&1 | grep " us " | tee -a $O/no_fragment_reads_after_the_first_group_b4.txt
done
