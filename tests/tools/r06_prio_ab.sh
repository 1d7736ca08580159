#!/bin/bash
# A/B on one box: priority time-slicing inside the 8-wave tiles (-DDD3D_ROW_PRIO_SLICE=1 / 2, csrc/conv_planes_row.hip) against the shipped library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in prio1 prio2; do
echo "== parity on $v"
DD3D_HIP_LIB=$R/build/ab/libdd3d_$v.so timeout 900 python -m pytest $R/tests/test_conv_planes_gpu.py "$R/tests/test_full_size_gpu.py::test_dla34_kitti_four_image_plan_matches_oracle" -q -m gpu -x 2>&1 | tail -1
done
DD3D_HIP_LIB=build/ab/libdd3d_prio1stamp.so timeout 300 python tests/gpu_row_stamp_probe.py 4 towers.1 level3.tree1.tree1.conv2 2>&1 | grep -v "amdgpu\|^\[build"
run() { DD3D_HIP_LIB=$2 timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
for rep in 1 2 3; do
run "shipped" ""
run "waves 4-7 first, then 0-3" $R/build/ab/libdd3d_prio1.so
run "waves 0-3 alternate 2/0" $R/build/ab/libdd3d_prio2.so
done
