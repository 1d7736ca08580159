#!/bin/bash
# The merged FPN output launch (3 segments, 40 320 rows at four images) on the 192 x 256 8-wave tile (210 blocks) against the 256 x 256 one (158 blocks on 256 CUs):
# parity of the new instantiation, the launch's in-graph cost, the driver command (alternating).
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest $R/tests/test_conv_planes_gpu.py -q -m gpu -x -k "192x256 or w8" 2>&1 | tail -2
K4="1920+7680+30720,256,2304,1"
for ov in "" "$K4=192x256w8:1"; do
  echo "== override '$ov'"
  DD3D_TILE_OVERRIDE="$ov" timeout 300 python $R/tests/gpu_prefix_bench.py 384 1280 4 2>&1 | grep "fpn_outputs\|nms_finalize"
done
run() { DD3D_TILE_OVERRIDE="$2" timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
for rep in 1 2 3; do
run "256x256w8" ""
run "192x256w8" "$K4=192x256w8:1"
done
