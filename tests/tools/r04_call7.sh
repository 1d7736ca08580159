#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest_gpu.txt
timeout 300 python tests/gpu_prefix_bench.py 384 1280 4 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b4.txt; tail -1 $O/prefix_b4.txt
timeout 300 python tests/gpu_prefix_bench.py 384 1280 1 2>&1 | grep -v "amdgpu\|build" > $O/prefix_b1.txt; tail -1 $O/prefix_b1.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
DD3D_PRED_SPLIT=0 timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pred1.json 2>> $O/bench.err; cut -c1-200 $O/bench_pred1.json
