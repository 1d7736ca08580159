R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o -E "\b(TCP|TCC|TA|TD|SQ|GRBM|SPI)_[A-Za-z0-9_]+" $O/counters_list.txt | sort -u > $O/counter_names.txt
wc -l $O/counter_names.txt
export DD3D_MATH=f16x2
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc/g$i -o out --output-format csv -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 > $O/pmc_g$i.log 2>&1
  echo "group $i rc=$?"
done
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT", ".")+"/gpurun_out/r02h"
agg=collections.defaultdict(list)
for f in glob.glob(O+"/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "planes_kernel<2, 2, 4, 2" in row.get("Kernel_Name",""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O+"/pmc_tower_f16x2.txt","w") as fo:
    for k,v in sorted(agg.items()):
        line=f"{k} {sum(v)/len(v):.1f} n={len(v)}"
        print(line); fo.write(line+"\n")
PY
