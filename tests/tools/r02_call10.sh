R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02j
mkdir -p $O
OPS=towers.1,level3.tree1.tree2.conv1,level4.tree2.tree1.conv1,fpn_output3
for v in base nomfma nodma nods dmaonly mfmaonly; do
  for m in f16x2 bf16x3; do
    DD3D_HIP_LIB=$R/build/ab/libdd3d_$v.so DD3D_MATH=$m timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v -E "amdgpu.ids|build" | sed "s/^/$v /" | tee -a $O/ablation.txt
  done
done
