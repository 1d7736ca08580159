#!/bin/bash
# SQ counters of EVERY convolution instantiation of the driver command's issue mode (5 slots x 4 images per launch plan, hipGraph replays):
# separate rocprofv3 --pmc passes over bench.py itself (<= 8 SQ counters each, --kernel-trace only), aggregated per (kernel, grid) and set
# against the launch plan's op table -- where the CU time of a forward goes, measured (round-5 verdict: the r05 argument rested on ISA counts).
#   bash tests/tools/r06_issue_pmc.sh <out.txt> [extra bench.py args]
OUT=${1:-gpurun_out/r06_issue_pmc.txt}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$R/gpurun_out/pmc_r06; rm -rf $T; mkdir -p $T
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $T/g$i -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --repeat-blocks 0 \
      --no-cpu-baseline --e2e-requests 0 --alt-issue '' "$@" > $T/bench_g$i.log 2>&1
done
python - <<PY > $R/$OUT
import csv, glob, collections, os, sys
sys.path.insert(0, "$R")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
meta = {}
for f in glob.glob("$T/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dd3d::", "")
        key = (k, int(row["Grid_Size"]), int(row["LDS_Block_Size"]))
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"], row["Workgroup_Size"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --kernel-trace --pmc (three separate passes) over: python bench.py --gpus 1 --steps 20 --warmup 5 --repeat-blocks 0 --no-cpu-baseline --e2e-requests 0 --alt-issue '' $*")
print("# per kernel instantiation, grid (threads) and LDS bytes: mean over ALL its launches of a pass (warm-up, timed block and the bench's probes alike);")
print("# SQ_*_CYCLES in quad-cycles summed over waves unless noted (MI355X_MICROARCH.md); under --pmc the profiler serialises the dispatches, so these are")
print("# the kernels' own counters, not the overlap of the five slots")
tot_cu = sum(sum(c.get("SQ_BUSY_CU_CYCLES", [0])) for c in agg.values()) or 1.0
rows = sorted(agg, key=lambda k: -sum(agg[k].get("SQ_BUSY_CU_CYCLES", [0])))
print(f"\n{'share of CU-busy':>17s} {'launches':>8s} {'mean us':>8s} {'MFMA busy':>9s} {'wait_inst':>9s} {'wait_any':>8s} {'issuing':>7s} {'LDS confl':>9s}  kernel, grid, lds")
for key in rows:
    c = agg[key]
    m = {nm: sum(x) / len(x) for nm, x in c.items()}
    n = len(c.get("SQ_BUSY_CU_CYCLES", next(iter(c.values()))))
    d = dur.get(key, [])
    share = sum(c.get("SQ_BUSY_CU_CYCLES", [0])) / tot_cu
    w = m.get("SQ_WAVE_CYCLES", 0) or 1.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * m["SQ_BUSY_CU_CYCLES"]) if m.get("SQ_BUSY_CU_CYCLES") else 0.0
    confl = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else 0.0
    print(f"{share:17.4f} {n:8d} {sum(d) / max(1, len(d)):8.1f} {busy:9.3f} {m.get('SQ_WAIT_INST_ANY', 0) / w:9.3f} {m.get('SQ_WAIT_ANY', 0) / w:8.3f} "
          f"{m.get('SQ_ACTIVE_INST_ANY', 0) / w:7.3f} {confl:9.4f}  {key[0]}  grid {key[1]}  lds {key[2]}")
print("\n# ---- every counter, per instantiation")
for key in rows:
    c = agg[key]
    v = meta[key]
    d = dur.get(key, [])
    print(f"\n{key[0]}  grid {key[1]} threads  mean duration under the profiler {sum(d) / max(1, len(d)):.1f} us (min {min(d):.1f} max {max(d):.1f})  "
          f"vgpr {v[0]} agpr {v[1]} sgpr {v[2]} lds {v[3]} scratch {v[4]} workgroup {v[5]}")
    m = {nm: sum(x) / len(x) for nm, x in c.items()}
    for nm in sorted(m):
        print(f"  {nm:28s} {m[nm]:16.0f}")
    ni = m.get("SQ_INSTS_MFMA")
    if ni:
        print("  per MFMA: " + "  ".join(f"{nm[9:]} {m[nm] / ni:.2f}" for nm in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM") if nm in m))
PY
for f in $T/bench_g*.log; do echo "--- $(basename $f)"; tail -2 $f | cut -c1-400; done >> $R/$OUT
mkdir -p $R/gpurun_out/pmc_r06_raw; find $T -name '*counter_collection.csv' | head -3 | xargs -I{} cp {} $R/gpurun_out/pmc_r06_raw/ 2>/dev/null; rm -rf $T
