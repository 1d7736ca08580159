R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02z4
mkdir -p $O
timeout 300 python -m pytest tests/test_nms_gpu.py tests/test_tta.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python -m pytest tests/test_forward_gpu.py -x -q -m gpu -k "per_class or v99 or nuscenes or bev" 2>&1 | tail -3
timeout 300 python tests/gpu_configs_check.py v99 1 2>&1 | grep dd3d_ | cut -c1-120 | tee $O/configs_v99.txt
timeout 300 python tests/gpu_configs_check.py nusc_dla34 2>&1 | grep dd3d_ | cut -c1-120 | tee -a $O/configs_v99.txt
