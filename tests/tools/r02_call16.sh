R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02p
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -12 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
bash tests/tools/r02_profile.sh r02b 2>&1 | head -40
