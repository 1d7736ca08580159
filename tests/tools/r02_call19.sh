R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02s
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -12 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tests/tools/r02_profile.sh r02c 2>&1 | head -30
timeout 300 python tests/gpu_prefix_bench.py > $R/gpurun_out/r02c/r02c_prefix.txt 2>&1
