R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/r06z_bench_kernel_stats.csv
python $R/tests/tools/kernel_stats_by_grid.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/r06z_bench_kernel_stats_by_grid.csv
head -8 $O/r06z_bench_kernel_stats_by_grid.csv | cut -c1-220
python -c "
import json; d=json.loads(open('$O/bench_prof.json').read().strip().splitlines()[-1]); print('live', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
rm -rf $O/prof
