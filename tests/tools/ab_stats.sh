#!/bin/bash
# A/B of env knobs on ONE box: kernel stats of the bench under each setting.   bash tests/tools/ab_stats.sh "A=1 B=2" "A=3" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab$i -o ab -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/ab$i.json 2>/dev/null
  echo "== $cfg"; grep -E "${AB_FILTER:-nms|select|bev}" $(find $R/gpurun_out/ab$i -name "*kernel_stats.csv") | awk -F, '{printf "%-70s %8.1f us\n", substr($1,1,70), $4/1000}'
done
