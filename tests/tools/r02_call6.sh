R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02f
mkdir -p $O
OPS=towers.1,towers.3,level3.tree1.tree2.conv1,level4.tree2.tree1.conv1,level5.tree2.conv1,fpn_output3,predictors
for v in 0 1 2; do
  for m in f16x2 bf16x3; do
    DD3D_HIP_LIB=$R/build/ab/libdd3d_v$v.so DD3D_MATH=$m timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v -E "amdgpu.ids|build" | sed "s/^/v$v /" | tee -a $O/variants.txt
  done
  DD3D_HIP_LIB=$R/build/ab/libdd3d_v$v.so DD3D_BENCH_TAG=v$v timeout 200 python tests/gpu_conv_bench.py 2>&1 | tail -2 | sed "s/^/v$v /" | tee -a $O/variants.txt
done
