R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 50 --warmup 10 > $R/gpurun_out/r01c_bench.json 2> $R/gpurun_out/r01c_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c_prof -o r01c -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r01c_bench_prof.json 2>/dev/null
cp $(find $R/gpurun_out/r01c_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01c_bench_kernel_stats.csv
cat $R/gpurun_out/r01c_bench.json; tail -2 $R/gpurun_out/r01c_bench.err; cat $R/gpurun_out/r01c_bench_prof.json
