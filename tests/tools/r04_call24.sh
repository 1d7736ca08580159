#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
for v in 1 0; do
  echo "== DD3D_TILE_ORDER=$v" | tee -a $O/tile_order.txt
  DD3D_TILE_ORDER=$v timeout 200 python tests/gpu_op_time.py 384 1280 4 towers fpn_outputs predictors 2>&1 | grep " us " | tee -a $O/tile_order.txt
done
timeout 600 python -m pytest tests/test_forward_gpu.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/tile_order.txt
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
DD3D_TILE_ORDER=$v timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('order=$v', d['value'], d['blocks']['ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/tile_order.txt
done
