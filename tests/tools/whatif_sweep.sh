#!/bin/bash
# In-situ decomposition of the driver command: throughput with parts of the forward replaced by no-ops (tests/gpu_whatif_bench.py).
#   bash tests/tools/whatif_sweep.sh <tag>  -> gpurun_out/<tag>_whatif_sweep.txt
TAG=${1:-r05n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
run() {
  DD3D_WHATIF_SKIP="$2" timeout 300 python tests/gpu_whatif_bench.py --gpus 1 --steps 20 --warmup 5 --repeat-blocks 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
c = d['config']
print('%-44s value %8.1f  median %8.1f  (%.4f ms/step)  one-slot %7.1f img/s  bs1 %7.1f img/s' % ('$1', d['value'], d['blocks']['median_images_per_s'], d['blocks']['median_ms_per_step'], c['images_per_s_one_slot_at_a_time'], c['bs1_images_per_s']))
" | tee -a $O/${TAG}_whatif_sweep.txt
}
run "everything (baseline)" ""
run "without the stem" "stem"
run "without level 2" "level2."
run "without levels 3-5" "level3.,level4.,level5."
run "without levels 2-5" "level2.,level3.,level4.,level5."
run "without FPN (laterals, outputs, p6/p7)" "fpn_,top_block"
run "without levels 2-5 and FPN" "level2.,level3.,level4.,level5.,fpn_,top_block"
run "without the towers" "towers."
run "without predictors + select + NMS" "predictors,select_decode,nms_finalize"
run "towers only" "stem,level2.,level3.,level4.,level5.,fpn_,top_block,predictors,select_decode,nms_finalize"
run "everything (baseline again)" ""
