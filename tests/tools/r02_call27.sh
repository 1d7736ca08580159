R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02z2
mkdir -p $O
timeout 300 python -m pytest tests/test_ese_gpu.py -x -q -m gpu 2>&1 | tail -6
for f in 1 0; do
  DD3D_ESE_FUSED=$f DD3D_EXP=dd3d_kitti_v99 timeout 300 python tests/gpu_prefix_bench.py > $O/prefix_v99_fused$f.txt 2>&1
  grep -E "ese|split|concat" $O/prefix_v99_fused$f.txt | head -12; tail -1 $O/prefix_v99_fused$f.txt
  DD3D_ESE_FUSED=$f timeout 300 python tests/gpu_configs_check.py kitti_v99 2>&1 | grep dd3d_ | cut -c1-110 | tee $O/configs_fused$f.txt
done
