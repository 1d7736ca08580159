#!/bin/bash
# In-situ A/B on top of the throughput tile policy: levels 4 / 5 (N = 256 / 512) on the towers' 8-wave 256 x 256 tile, no split-K (30 / 16 blocks
# per launch -- only sensible when other slots fill the chip).   bash tests/tools/r06_bm256_sweep2.sh [reps]
run() { DD3D_TILE_POLICY=$3 DD3D_TILE_OVERRIDE="$2" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
L4="7680,256,2304,1"; L5="1920,512,4608,1"
for rep in $(seq 1 ${1:-3}); do
run throughput-table "" throughput
run "+level4 256x256w8" "$L4=256x256w8:1" throughput
run "+level4,5 256x256w8" "$L4=256x256w8:1;$L5=256x256w8:1" throughput
done
