R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02z3
mkdir -p $O
timeout 300 python -m pytest tests/test_ese_gpu.py -x -q -m gpu 2>&1 | tail -4
DD3D_EXP=dd3d_kitti_v99 timeout 300 python tests/gpu_prefix_bench.py > $O/prefix_v99_fused1.txt 2>&1
grep -E "\.ese" $O/prefix_v99_fused1.txt | awk '{printf "%s %s | ", $1, $3}'; echo; tail -1 $O/prefix_v99_fused1.txt
for f in 1 0; do
  DD3D_ESE_FUSED=$f timeout 300 python tests/gpu_configs_check.py kitti_v99 2>&1 | grep dd3d_ | cut -c1-110 | tee $O/configs_fused$f.txt
done
