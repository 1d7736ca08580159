R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02e
mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $O/r02e_pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/bench.py --steps 50 --warmup 10 --repeat-blocks 2 > $O/r02e_bench.json 2>/dev/null; cut -c1-330 $O/r02e_bench.json
