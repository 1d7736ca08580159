R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02m
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_planes_gpu.py tests/test_parallel_gpu.py tests/test_forward_gpu.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
timeout 300 python tests/gpu_dist_check.py 2 2>&1 | tail -2
bash tests/tools/r02_profile.sh r02a
