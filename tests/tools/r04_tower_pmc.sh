#!/bin/bash
# SQ counters of the shipped head-tower launch (conv_igemm_planes_row_kernel<4,2,2,4,2,4,false>, f16x2, B images per launch) and of one
# backbone convolution: separate rocprofv3 --pmc passes (<= 8 SQ counters each), --kernel-trace only.  bash tests/tools/r04_tower_pmc.sh <B> <out.txt>
B=${1:-4}; OUT=${2:-gpurun_out/r04_tower_f16x2_pmc.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$R/gpurun_out/pmc_r04; rm -rf $T; mkdir -p $T
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $T/g$i -o p -- python $R/tests/gpu_pmc_probe.py towers.1,level3.tree1.tree1.conv2,level2.tree2.conv2 3 $B > /dev/null 2>&1
done
python - <<PY > $R/$OUT
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
meta = {}
for f in glob.glob("$T/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dd3d::", "")
        if "conv_igemm_planes" not in k:
            continue
        key = (k, int(row["Grid_Size"]))
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"], row["Workgroup_Size"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --kernel-trace --pmc (three separate passes), tests/gpu_pmc_probe.py towers.1,level3.tree1.tree1.conv2,level2.tree2.conv2 3 $B (f16x2, $B images per launch)")
print("# per kernel instantiation and grid: mean over the launches of the pass; SQ_*_CYCLES in quad-cycles summed over waves unless noted (MI355X_MICROARCH.md)")
for key in sorted(agg, key=lambda k: -k[1]):
    c = agg[key]
    n = len(next(iter(c.values())))
    d = dur.get(key, [])
    v = meta[key]
    print(f"\n{key[0]}  grid {key[1]} threads  launches/pass {n}  mean duration under the profiler {sum(d) / max(1, len(d)):.1f} us  "
          f"vgpr {v[0]} agpr {v[1]} sgpr {v[2]} lds {v[3]} scratch {v[4]} workgroup {v[5]}")
    m = {nm: sum(x) / len(x) for nm, x in c.items()}
    for nm in sorted(m):
        print(f"  {nm:28s} {m[nm]:16.0f}")
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if nm in m: print(f"  {nm} / SQ_WAVE_CYCLES = {m[nm] / w:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
        print(f"  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.3f}")
PY
mkdir -p $R/gpurun_out/pmc_r04_raw; find $T -name '*counter_collection.csv' | head -3 | xargs -I{} cp {} $R/gpurun_out/pmc_r04_raw/ 2>/dev/null; rm -rf $T
