#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04r
mkdir -p $O
cd $R
for v in "" setprio; do
  echo "== ${v:-shipped}" | tee -a $O/op_time_b4.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 level2.tree2 level3.tree1.tree1.conv2 level5.tree1.conv2 fpn_outputs towers 2>&1 | grep " us " | tee -a $O/op_time_b4.txt
done
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
for v in "" setprio; do
DD3D_HIP_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-new}', d['value'], d['blocks']['ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/bench_ab.txt
done; done
