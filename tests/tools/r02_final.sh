#!/bin/bash
# End-of-round check: the whole GPU suite, smoke, the bench lines with their rocprofv3 kernel stats, per-op in-graph costs, the other
# BASELINE configurations.   bash tests/tools/r02_final.sh r02d
TAG=${1:-r02d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/${TAG}_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps 50 --warmup 10 > $O/${TAG}_bench.json 2> $O/bench.err; cat $O/${TAG}_bench.json | cut -c1-600
timeout 400 python $R/bench.py --steps 50 --warmup 10 --pipeline 0 --no-cpu-baseline > $O/${TAG}_bench_serial.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/${TAG}_bench_serial.json')); print('serial', d['value'], d['blocks']['median_images_per_s'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python $R/bench.py --steps 50 --warmup 10 --pipeline 0 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof_serial.json 2>/dev/null
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_serial_kernel_stats.csv
head -6 $O/${TAG}_bench_serial_kernel_stats.csv | cut -c1-200
rm -rf $O/prof $O/prof_serial
cd $R
timeout 200 python tests/gpu_prefix_bench.py 2>&1 | grep -v "amdgpu\|build" > $O/${TAG}_prefix.txt; tail -3 $O/${TAG}_prefix.txt
timeout 600 python tests/gpu_configs_check.py 2>&1 | grep dd3d_ | cut -c1-200 | tee $O/${TAG}_configs.txt
