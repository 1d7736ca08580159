R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02q
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_planes_gpu.py -x -q > $O/pytest_planes.log 2>&1; echo "planes rc=$?"; tail -6 $O/pytest_planes.log
OPS=towers.1,towers.3,level3.tree1.tree2.conv1,level4.tree2.tree1.conv1,level5.tree2.conv1,fpn_output3,predictors
for row in 1 0; do
  for m in f16x2 bf16x3; do
    DD3D_CONV_ROW=$row DD3D_MATH=$m timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v -E "amdgpu.ids|build" | sed "s/^/row=$row /" | tee -a $O/row.txt
  done
done
for row in 1 0 1 0; do
DD3D_CONV_ROW=$row timeout 300 python bench.py --steps 60 --warmup 16 --no-cpu-baseline --repeat-blocks 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('row=$row', 'pipelined', d['value'], 'median', d['blocks']['median_images_per_s'], '| serial', c['images_per_s_one_at_a_time'], '| tower us', d['roofline']['avg_launch_us'], d['roofline']['kernel'][:70])" | tee -a $O/row.txt
done
timeout 600 python -m pytest tests/test_forward_gpu.py tests/test_full_size_gpu.py -x -q -k "not nuscenes_sample and not bs16" 2>&1 | tail -3
