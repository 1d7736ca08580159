R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02g
mkdir -p $O
for ws in 96 512 2048; do echo "== working set per block $ws KiB"; timeout 120 ./build/bin/lds_dma_bench $ws; done 2>&1 | tee $O/lds_dma_bench2.txt
