R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02o
mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 80 --warmup 16 --no-cpu-baseline --repeat-blocks 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$tag', 'value', d['value'], 'median', d['blocks']['median_images_per_s'], 'min/max ms', d['blocks']['min_ms_per_step'], d['blocks']['max_ms_per_step'], '| serial', c['images_per_s_one_at_a_time'])" | tee -a $O/depth.txt; }
for lib in b72 b48; do
  for d in 4 8 12 16; do
    run "$lib depth $d" DD3D_HIP_LIB=$R/build/ab/libdd3d_$lib.so DD3D_BENCH_PIPELINE=$d DD3D_BENCH_COMPUTE_STREAMS=$d
  done
done
run "b72 depth 16 streams 8" DD3D_HIP_LIB=$R/build/ab/libdd3d_b72.so DD3D_BENCH_PIPELINE=16 DD3D_BENCH_COMPUTE_STREAMS=8
