#!/bin/bash
# Round-2 profile: bench lines, rocprofv3 kernel stats of the same commands, PMC HBM-traffic passes (separate passes, never with
# trace domains) for the dominant kernels, the other BASELINE configurations.   bash tests/tools/r02_profile.sh r02a
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps 50 --warmup 10 > $O/${TAG}_bench.json 2> $O/bench.err; cat $O/${TAG}_bench.json
timeout 400 python $R/bench.py --steps 50 --warmup 10 --pipeline 0 --no-cpu-baseline > $O/${TAG}_bench_serial.json 2>> $O/bench.err
timeout 400 python $R/bench.py --steps 50 --warmup 10 --math bf16x3 --no-cpu-baseline > $O/${TAG}_bench_bf16x3.json 2>> $O/bench.err
DD3D_BENCH_PIPELINE=8 DD3D_BENCH_COMPUTE_STREAMS=8 timeout 400 python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_p8.json 2>> $O/bench.err
for f in serial bf16x3 p8; do python -c "
import json; d=json.load(open('$O/${TAG}_bench_$f.json')); print('$f', d['value'], d['blocks']['median_images_per_s'], d['config']['images_per_s_one_at_a_time'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof.json 2>/dev/null
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python $R/bench.py --steps 50 --warmup 10 --pipeline 0 --no-cpu-baseline --repeat-blocks 0 > $O/bench_prof_serial.json 2>/dev/null
cp $(find $O/prof_serial -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_serial_kernel_stats.csv
head -8 $O/${TAG}_bench_serial_kernel_stats.csv
# HBM traffic of the tower kernel: FETCH_SIZE and WRITE_SIZE in separate passes
for m in f16x2 bf16x3; do
  DD3D_MATH=$m timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$m -o f -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 > /dev/null 2>&1
  DD3D_MATH=$m timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write_$m -o w -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for m in ("f16x2", "bf16x3"):
    agg = collections.defaultdict(list)
    names = set()
    for f in glob.glob("$O/pmc_*_%s/**/*counter_collection.csv" % m, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "conv_igemm_planes_row_kernel<2, 2, 4, 2" in k or "conv_igemm_planes_kernel<2, 2, 4, 2" in k:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                names.add(k.split("(")[0])
    out[m] = {"kernel_names": sorted(names), "counters_avg_per_launch": {k: sum(v) / len(v) for k, v in agg.items()}, "launches": {k: len(v) for k, v in agg.items()}}
json.dump(out, open("$O/${TAG}_tower_pmc_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
cd $R
for m in f16x2 bf16; do DD3D_MATH=$m timeout 900 python tests/gpu_configs_check.py 2>&1 | grep -v amdgpu | tee -a $O/${TAG}_configs.txt; done
