#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
timeout 300 python tests/gpu_rccl_check.py 2>&1 | grep "step (\|rccl check" | tee -a $O/exchange_modes.txt
timeout 300 python tests/gpu_rccl_check.py graph 2>&1 | grep "step (\|rccl check" | tee -a $O/exchange_modes.txt
