#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats of the same command, PMC HBM-traffic passes for the tower kernel.
#   bash tests/tools/profile_round.sh r01c        (on the GPU box; writes gpurun_out/<tag>_*)
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 50 --warmup 10 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.json 2>/dev/null
cp $(find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python $R/tests/gpu_pmc_probe.py towers.1 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python $R/tests/gpu_pmc_probe.py towers.1 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/${TAG}_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "conv_igemm" in k and int(row["Grid_Size"]) in (252 * 512, 486 * 256):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(json.dumps({k: [sum(v) / len(v), len(v)] for k, v in agg.items()}))
PY
cat $R/gpurun_out/${TAG}_bench.json
head -12 $R/gpurun_out/${TAG}_bench_kernel_stats.csv
