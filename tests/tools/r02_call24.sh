R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02x
mkdir -p $O
timeout 400 python -m pytest tests/test_nms_gpu.py tests/test_ese_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_forward_gpu.py tests/test_tta.py tests/test_zz_empty_inputs.py -x -q -m gpu 2>&1 | tail -8
timeout 200 python tests/gpu_prefix_bench.py > $O/prefix.txt 2>&1; tail -4 $O/prefix.txt
