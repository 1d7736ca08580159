#!/bin/bash
# End-of-round artifacts: bench line of the default command, rocprofv3 kernel stats of the same command and of the one-step-at-a-time
# issue mode (whose per-kernel durations are the isolated ones the roofline object quotes).
#   bash tests/tools/final_round.sh r01f        (on the GPU box; writes gpurun_out/<tag>_*)
TAG=${1:-r01f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 120 python $R/bench.py --steps 100 --warmup 20 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.json 2>/dev/null
cp $(find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof1 -o ${TAG} -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --pipeline 0 > $R/gpurun_out/${TAG}_bench_serial_prof.json 2>/dev/null
cp $(find $R/gpurun_out/${TAG}_prof1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_serial_kernel_stats.csv
cat $R/gpurun_out/${TAG}_bench.json; tail -2 $R/gpurun_out/${TAG}_bench.err
head -4 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-200
head -4 $R/gpurun_out/${TAG}_bench_serial_kernel_stats.csv | cut -c1-200
cut -c1-200 $R/gpurun_out/${TAG}_bench_serial_prof.json
