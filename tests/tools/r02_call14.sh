R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02n
mkdir -p $O
DD3D_HIP_LIB=$R/build/ab/libdd3d_pw4.so timeout 600 python -m pytest tests/test_conv_planes_gpu.py -x -q 2>&1 | tail -3
OPS=towers.1,towers.3,fpn_output3
for v in pw0 pw2 pw4; do
  for m in f16x2 bf16x3; do
    DD3D_HIP_LIB=$R/build/ab/libdd3d_$v.so DD3D_MATH=$m timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v -E "amdgpu.ids|build" | sed "s/^/$v /" | tee -a $O/variants.txt
  done
done
for v in pw0 pw4 pw0 pw4; do
DD3D_HIP_LIB=$R/build/ab/libdd3d_$v.so timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --repeat-blocks 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$v', 'pipelined', d['value'], 'median', d['blocks']['median_images_per_s'], '| serial', c['images_per_s_one_at_a_time'], '| tower us', d['roofline']['avg_launch_us'])" | tee -a $O/variants.txt
done
DD3D_EXP=x DD3D_HIP_LIB=$R/build/ab/libdd3d_pw4.so DD3D_MATH=f16x2 timeout 600 python tests/gpu_configs_check.py 2>&1 | grep -v amdgpu | head -4 | sed "s/^/pw4 /" | tee -a $O/variants.txt
