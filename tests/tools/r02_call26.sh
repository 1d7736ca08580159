R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02z
mkdir -p $O
timeout 300 python -m pytest tests/test_ese_gpu.py -x -q -m gpu 2>&1 | tail -12
timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -m gpu -k "v99 or golden" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_zz_backbone_variants.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python tests/gpu_configs_check.py v99 > $O/configs_v99.txt 2>&1; cut -c1-130 $O/configs_v99.txt | tail -4
