# round-2 GPU call 2: what bounds the tower kernel (power-limited clock vs stalls), and what the reduced modes give
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02b
mkdir -p $O
OPS=towers.1,towers.3,level3.tree1.tree2.conv1,level4.tree2.tree1.conv1,fpn_output3
for m in bf16x3 bf16x2 bf16; do DD3D_MATH=$m timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v amdgpu.ids | tee -a $O/tower_probe.txt; done
DD3D_PLANES=0 timeout 200 python tests/gpu_tower_probe.py $OPS 2>&1 | grep -v amdgpu.ids | tee -a $O/tower_probe.txt
for m in bf16x2 bf16; do DD3D_MATH=$m DD3D_BENCH_TAG=$m timeout 200 python tests/gpu_conv_bench.py > $O/conv_bench_$m.txt 2>&1; tail -2 $O/conv_bench_$m.txt; done
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc/$tag -o out --output-format csv -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT", ".")+"/gpurun_out/r02b"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "planes" in row.get("Kernel_Name","") and "split" not in row.get("Kernel_Name",""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O+"/pmc_summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        line=f"{k} {sum(v)/len(v):.1f} n={len(v)}"
        print(line); fo.write(line+"\n")
PY
