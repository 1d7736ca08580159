#!/bin/bash
# rocprofv3 kernel stats + HBM counters (separate passes) of DD3D-V2-99 KITTI 384x1280 bs=16 (BASELINE.json configs[2]).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tests/gpu_kernel_traffic.py run dd3d_kitti_v99 16 384 1280 3 $O/plan.json"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/run.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $CMD >> $O/run.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -o w -- $CMD >> $O/run.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r02_v99_bs16_kernel_stats.csv
python $R/tests/gpu_kernel_traffic.py report $O/plan.json $O/r02_v99_bs16_kernel_stats.csv $O/fetch $O/write $O/r02_v99_bs16_traffic.json
head -5 $O/r02_v99_bs16_kernel_stats.csv | cut -c1-180
rm -rf $O/stats $O/fetch $O/write
tail -3 $O/run.log
