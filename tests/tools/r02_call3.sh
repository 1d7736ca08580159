R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02c
mkdir -p $O
timeout 600 python -m pytest tests/test_conv_planes_gpu.py -x -q > $O/pytest_planes.log 2>&1; echo "planes rc=$?"
tail -5 $O/pytest_planes.log
DD3D_MATH=f16x2 timeout 200 python tests/gpu_tower_probe.py towers.1,towers.3,level3.tree1.tree2.conv1,fpn_output3 2>&1 | grep -v amdgpu.ids | tee $O/tower_probe_f16x2.txt
timeout 900 python tests/gpu_math_modes.py both > $O/math_modes.txt 2>&1; grep -v amdgpu.ids $O/math_modes.txt | tail -20
cp gpurun_out/math_modes.json $O/
