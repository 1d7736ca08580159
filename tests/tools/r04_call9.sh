#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_planes_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest_conv.txt
for v in "" nsa2; do
  echo "== ${v:-shipped (NSA up to 4)}" | tee -a $O/op_time_b4.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 level2.tree level3.tree level4.tree1.tree level5.tree fpn_outputs towers.0 predictors 2>&1 | grep " us " | tee -a $O/op_time_b4.txt
done
for v in "" nsa2; do
  echo "== ${v:-shipped (NSA up to 4)}" | tee -a $O/op_time_b1.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 1 level2.tree level3.tree level4.tree1.tree level5.tree fpn_outputs towers.0 predictors 2>&1 | grep " us " | tee -a $O/op_time_b1.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
DD3D_HIP_LIB=$R/build/ab/libdd3d_nsa2.so timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nsa2.json 2>> $O/bench.err; cut -c1-200 $O/bench_nsa2.json
