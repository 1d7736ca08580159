#!/bin/bash
# HBM traffic of the head-tower launch of the DD3D-DLA34 plan for B images (the kernel bench.py's `roofline` is about), per
# /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, never combined
# with trace domains other than --kernel-trace.   bash tests/tools/tower_traffic.sh <B> <out.json>
B=${1:-5}; OUT=${2:-gpurun_out/tower_hbm_bytes.json}
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=$R/gpurun_out/tower_pmc_b$B; rm -rf $T; mkdir -p $T
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -o f -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $T/write -o w -- python $R/tests/gpu_pmc_probe.py towers.1,towers.2 4 $B > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json, os
# the tower launches: the forward that fills the buffers runs towers.0 .. 3 once, the probe towers.1 and towers.2 four times each -> the
# (kernel, grid) pair that was launched exactly 12 times with the largest grid
groups = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$T/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_igemm_planes" in k:
            groups[(k, int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
cands = [(g, k, c) for (k, g), c in groups.items() if len(c.get("FETCH_SIZE", [])) == 12 and len(c.get("WRITE_SIZE", [])) == 12]
out = json.load(open("$R/$OUT")) if os.path.exists("$R/$OUT") else {}
for g, k, c in sorted(cands)[-1:]:
    k = f"{k} @ {$B} images per launch"
    fetch, write = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 2 * 1024, sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    h, m = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
    out[k] = {"images_per_launch": $B, "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
              "tcc_hit_rate": round(h / (h + m), 4) if h + m else None, "launches_averaged": len(c["FETCH_SIZE"]), "grid_threads": g,
              "collected": "round 6, tests/tools/tower_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (own pass) and --pmc WRITE_SIZE TCC_HIT_sum "
                           "TCC_MISS_sum (own pass) over tests/gpu_pmc_probe.py towers.1,towers.2 4 $B; FETCH_SIZE x 2 x 1024 B (gfx950 reports half the bytes "
                           "of wide coalesced reads, MI355X_MICROARCH.md section HBM), WRITE_SIZE x 1024 B"}
json.dump(out, open("$R/$OUT", "w"), indent=1)
print(json.dumps(out)[:1200])
PY
rm -rf $T
