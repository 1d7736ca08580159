#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04zz
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for q in 6 4 8 6 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeat-blocks 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq=$q', d['value'], d['blocks']['ms_per_step'])" | tee -a $O/hw_queues.txt
done
