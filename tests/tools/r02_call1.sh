# round-2 GPU call 1: correctness of the split-plane conv path, then A/B against the round-1 data flow
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02a
mkdir -p $O
timeout 600 python -m pytest tests/test_conv_planes_gpu.py -x -q > $O/pytest_planes.log 2>&1; echo "planes rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_planes.log
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q > $O/pytest_conv.log 2>&1; echo "conv rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_conv.log
timeout 900 python -m pytest tests/test_forward_gpu.py -x -q -s > $O/pytest_forward.log 2>&1; echo "forward rc=$?" | tee -a $O/summary.txt
grep -E "margin|passed|failed|Error" $O/pytest_forward.log | tail -15
DD3D_BENCH_TAG=planes timeout 300 python tests/gpu_conv_bench.py > $O/conv_bench_planes.txt 2>&1; tail -4 $O/conv_bench_planes.txt
DD3D_PLANES=0 DD3D_BENCH_TAG=f32in timeout 300 python tests/gpu_conv_bench.py > $O/conv_bench_f32in.txt 2>&1; tail -4 $O/conv_bench_f32in.txt
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_planes.json 2> $O/bench_planes.err; cat $O/bench_planes.json
DD3D_PLANES=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_f32in.json 2> $O/bench_f32in.err; cat $O/bench_f32in.json
