R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== parity on the shipped library (scalar-base filter pieces)"
timeout 900 python -m pytest $R/tests/test_conv_planes_gpu.py $R/tests/test_chain_gpu.py $R/tests/test_conv_gpu.py "$R/tests/test_full_size_gpu.py::test_dla34_kitti_four_image_plan_matches_oracle" -q -m gpu -x 2>&1 | tail -2
run() { DD3D_HIP_LIB=$2 timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['blocks']['median_images_per_s'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'bs1', d['config']['bs1_ms_per_image'], 'slot alone', d['config']['ms_per_step_one_slot_at_a_time'])
"; }
for rep in 1 2 3; do
run "scalar-base (shipped)" ""
run "64-bit lane addresses" $R/build/ab/libdd3d_nosaddr.so
done | tee gpurun_out/r06o_saddr_ab2.txt
