#!/bin/bash
# A/B on one box (alternating): the shipped library vs variants of the row kernel's filter-stage issue --
#   saddr: filter pieces by the scalar-base LDS-DMA form (-DDD3D_ROW_B_SADDR=1)
#   bw4:   + only the first four waves of an 8-wave block issue them (-DDD3D_ROW_B_WAVES=4)
# Parity first (the convolution unit tests + the 4-image plan against the oracle on each variant), then the driver command.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in saddr bw4; do
  echo "== parity on $v"
  DD3D_HIP_LIB=$R/build/ab/libdd3d_$v.so timeout 900 python -m pytest $R/tests/test_conv_planes_gpu.py $R/tests/test_chain_gpu.py "$R/tests/test_full_size_gpu.py::test_dla34_kitti_four_image_plan_matches_oracle" -q -m gpu -x 2>&1 | tail -2
done
run() { DD3D_HIP_LIB=$2 timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
for rep in $(seq 1 ${1:-3}); do
run shipped ""
run saddr $R/build/ab/libdd3d_saddr.so
run bw4 $R/build/ab/libdd3d_bw4.so
done
