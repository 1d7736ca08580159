R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02k
mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --repeat-blocks 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$tag', 'pipelined', d['value'], 'img/s median', d['blocks']['median_images_per_s'], '| serial', c['images_per_s_one_at_a_time'], '| tower us', d['roofline']['avg_launch_us'])" | tee -a $O/policy.txt; }
for lib in l144_72 l96_48 l144_48; do
  run "$lib default" DD3D_HIP_LIB=$R/build/ab/libdd3d_$lib.so
  run "$lib thin   " DD3D_HIP_LIB=$R/build/ab/libdd3d_$lib.so DD3D_TILE_POLICY=thin
done
run "l96_48 thin 8/8" DD3D_HIP_LIB=$R/build/ab/libdd3d_l96_48.so DD3D_TILE_POLICY=thin DD3D_BENCH_PIPELINE=8 DD3D_BENCH_COMPUTE_STREAMS=8
run "l96_48 thin 6/6" DD3D_HIP_LIB=$R/build/ab/libdd3d_l96_48.so DD3D_TILE_POLICY=thin DD3D_BENCH_PIPELINE=6 DD3D_BENCH_COMPUTE_STREAMS=6
run "l144_72 default 8/8" DD3D_HIP_LIB=$R/build/ab/libdd3d_l144_72.so DD3D_BENCH_PIPELINE=8 DD3D_BENCH_COMPUTE_STREAMS=8
