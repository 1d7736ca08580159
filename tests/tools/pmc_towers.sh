cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DD3D_MATH=bf16x3
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_x3/$tag -o out --output-format csv -- python $R/tests/gpu_pmc_probe.py towers.1 3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/pmc_x3/*/*counter_collection.csv")+glob.glob(R+"/gpurun_out/pmc_x3/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "bf16x3" in row.get("Kernel_Name",""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(k, sum(v)/len(v), len(v))
PY
