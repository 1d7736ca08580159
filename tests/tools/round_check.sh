#!/bin/bash
# End-of-round check on the GPU box: the whole GPU suite, smoke, the driver's bench command, its rocprofv3 kernel stats, per-op in-graph costs,
# the other BASELINE configurations.   bash tests/tools/round_check.sh <tag>      (writes gpurun_out/<tag>/<tag>_*)
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/${TAG}_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_driver_cmd.json 2> $O/bench.err; cut -c1-300 $O/${TAG}_driver_cmd.json
python -c "
import json; d=json.load(open('$O/${TAG}_driver_cmd.json')); print('driver', d['value'], d['blocks']['ms_per_step'], d['blocks']['block0_over_median'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['cpu_baseline'])"
timeout 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline > $O/${TAG}_bench_serial.json 2>> $O/bench.err
python -c "
import json; d=json.load(open('$O/${TAG}_bench_serial.json')); print('one image at a time', d['value'], d['blocks']['median_images_per_s'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
python $R/tests/tools/kernel_stats_by_grid.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/${TAG}_bench_kernel_stats_by_grid.csv; head -4 $O/${TAG}_bench_kernel_stats_by_grid.csv | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --pipeline 0 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof_serial.json 2>/dev/null
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_serial_kernel_stats.csv
head -6 $O/${TAG}_bench_kernel_stats.csv | cut -c1-200; head -4 $O/${TAG}_bench_serial_kernel_stats.csv | cut -c1-200
rm -rf $O/prof $O/prof_serial
cd $R
timeout 200 python tests/gpu_prefix_bench.py 2>&1 | grep -v "amdgpu\|build" > $O/${TAG}_prefix_b1.txt; tail -3 $O/${TAG}_prefix_b1.txt
timeout 300 python tests/gpu_prefix_bench.py 384 1280 4 2>&1 | grep -v "amdgpu\|build" > $O/${TAG}_prefix_b4.txt; tail -3 $O/${TAG}_prefix_b4.txt
bash tests/tools/tower_traffic.sh 4 gpurun_out/$TAG/${TAG}_tower_hbm_bytes.json | cut -c1-400
bash tests/tools/tower_traffic.sh 1 gpurun_out/$TAG/${TAG}_tower_hbm_bytes.json | cut -c1-400
timeout 900 python tests/gpu_configs_check.py 2>&1 | grep dd3d_ | cut -c1-200 | tee $O/${TAG}_configs.txt
