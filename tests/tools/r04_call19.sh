#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04s
mkdir -p $O
cd $R
timeout 300 python tests/gpu_stage_check.py 2>&1 | tail -6 | tee $O/stage_check.txt
timeout 300 python tests/gpu_conv_bench.py 384 1280 4 2>&1 | tail -4 | tee $O/conv_bench_tail.txt
timeout 600 python tests/gpu_math_modes.py 2>&1 | tail -12 | tee $O/math_modes.txt
timeout 300 python tests/gpu_pipeline_check.py microbatch 2>&1 | tail -2
