#!/bin/bash
# In-situ A/B (the driver command, alternating): launch plans tiled for one launch's latency (the measured per-launch table) vs for CU-time
# (engine.tiling.THROUGHPUT_TILE_TABLE: the backbone's short-K 3 x 3 convolutions on 256 x 128 tiles with little split-K -- half the filter bytes
# through LDS per MFMA), plus single overrides on top.   bash tests/tools/r06_bm256_sweep.sh [reps]
run() { DD3D_TILE_POLICY=$3 DD3D_TILE_OVERRIDE="$2" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
L3s="30720,128,576,2"; L4s="7680,256,1152,2"; L5s="1920,512,2304,2"
for rep in $(seq 1 ${1:-3}); do
run latency-table "" latency
run throughput-table "" throughput
run "+stride-2 convs 256x128" "$L3s=256x128:1;$L4s=256x128:2;$L5s=256x128:4" throughput
done
