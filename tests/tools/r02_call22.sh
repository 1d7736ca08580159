R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02v
mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 80 --warmup 16 --no-cpu-baseline --repeat-blocks 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$tag', 'value', d['value'], 'median', d['blocks']['median_images_per_s'], '| serial', c['images_per_s_one_at_a_time'], '| tower', d['roofline']['avg_launch_us'])" | tee -a $O/queues.txt; }
run "default" X=1
run "hwq8" GPU_MAX_HW_QUEUES=8
run "hwq16" GPU_MAX_HW_QUEUES=16
run "hwq2" GPU_MAX_HW_QUEUES=2
run "hwq16 depth32" GPU_MAX_HW_QUEUES=16 DD3D_BENCH_PIPELINE=32 DD3D_BENCH_COMPUTE_STREAMS=32
run "default depth32" DD3D_BENCH_PIPELINE=32 DD3D_BENCH_COMPUTE_STREAMS=32
