R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02t
mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 80 --warmup 16 --no-cpu-baseline --repeat-blocks 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$tag', 'value', d['value'], 'median', d['blocks']['median_images_per_s'], '| serial', c['images_per_s_one_at_a_time'], '| tower', d['roofline']['avg_launch_us'])" | tee -a $O/lds.txt; }
for lib in base s40 t104s40 base s40; do
  run "$lib" DD3D_HIP_LIB=$R/build/ab/libdd3d_$lib.so
done
