#!/bin/bash
# Round 4, first GPU call: parity of the transposed epilogue / planes-only data flow, then in-graph per-op costs against round 3's forms.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_planes_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_conv_planes.txt
timeout 600 python -m pytest tests/test_forward_gpu.py tests/test_full_size_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_forward.txt
for B in 1 4; do
  timeout 300 python tests/gpu_prefix_bench.py 384 1280 $B 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b${B}_new.txt; tail -1 $O/prefix_b${B}_new.txt
  DD3D_PLANES_ONLY=0 timeout 300 python tests/gpu_prefix_bench.py 384 1280 $B 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b${B}_epiT_twins.txt; tail -1 $O/prefix_b${B}_epiT_twins.txt
  DD3D_PLANES_ONLY=0 DD3D_HIP_LIB=$R/build/ab/libdd3d_epi0.so timeout 300 python tests/gpu_prefix_bench.py 384 1280 $B 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b${B}_r03form.txt; tail -1 $O/prefix_b${B}_r03form.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_new.json 2> $O/bench.err; cut -c1-200 $O/bench_new.json
DD3D_PLANES_ONLY=0 DD3D_HIP_LIB=$R/build/ab/libdd3d_epi0.so timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r03form.json 2>> $O/bench.err; cut -c1-200 $O/bench_r03form.json
