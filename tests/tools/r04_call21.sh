#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04u
mkdir -p $O
cd $R
timeout 900 python tests/gpu_configs_check.py 2>&1 | grep dd3d_ | cut -c1-200 | tee $O/configs.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_forward_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['blocks']['ms_per_step'], d['roofline']['avg_launch_us'])" | tee $O/bench.txt
