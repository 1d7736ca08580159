#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_planes_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest_conv.txt
timeout 600 python -m pytest tests/test_forward_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest_forward.txt
export DD3D_AMAX=1
for v in "" epilds0; do
  echo "== ${v:-shipped (LDS-staged plane stores)}" | tee -a $O/op_time_b4.txt
  DD3D_HIP_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 2>&1 | grep " us " | tee -a $O/op_time_b4.txt
done
timeout 300 python tests/gpu_prefix_bench.py 384 1280 4 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b4.txt; tail -1 $O/prefix_b4.txt
timeout 300 python tests/gpu_prefix_bench.py 384 1280 1 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b1.txt; tail -1 $O/prefix_b1.txt
DD3D_BRANCHES=1 timeout 300 python tests/gpu_prefix_bench.py 384 1280 1 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b1_branches.txt; tail -1 $O/prefix_b1_branches.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
DD3D_HIP_LIB=$R/build/ab/libdd3d_epilds0.so timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_epilds0.json 2>> $O/bench.err; cut -c1-200 $O/bench_epilds0.json
DD3D_BRANCHES=1 timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_branches.json 2>> $O/bench.err; cut -c1-200 $O/bench_branches.json
tail -5 $O/bench.err
