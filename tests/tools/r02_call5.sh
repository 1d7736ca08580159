R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02e
mkdir -p $O
timeout 120 ./build/bin/lds_dma_bench > $O/lds_dma_bench.txt 2>&1; cat $O/lds_dma_bench.txt
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -16 $O/pytest_gpu.log
