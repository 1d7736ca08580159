R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02d
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -25 $O/pytest_gpu.log
timeout 400 python bench.py --steps 50 --warmup 10 > $O/bench_f16x2.json 2> $O/bench_f16x2.err; cat $O/bench_f16x2.json; tail -3 $O/bench_f16x2.err
timeout 400 python bench.py --steps 50 --warmup 10 --math bf16x3 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; cat $O/bench_bf16x3.json; tail -3 $O/bench_bf16x3.err
DD3D_BENCH_TAG=f16x2 timeout 300 python tests/gpu_conv_bench.py > $O/conv_bench_f16x2.txt 2>&1; tail -3 $O/conv_bench_f16x2.txt
timeout 300 python tests/gpu_prefix_bench.py > $O/prefix_f16x2.txt 2>&1; tail -5 $O/prefix_f16x2.txt
