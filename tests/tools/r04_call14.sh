#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial w5', d['value'], d['blocks']['ms_per_step'])" | tee -a $O/serial.txt
done
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 40 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial w40', d['value'], d['blocks']['ms_per_step'])" | tee -a $O/serial.txt
DD3D_PLANES_ONLY=0 timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial w5 twins', d['value'], d['blocks']['ms_per_step'])" | tee -a $O/serial.txt
