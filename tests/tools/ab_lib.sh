#!/bin/bash
# A/B on ONE box: working-tree library vs dd3d_amd/lib/libdd3d_hip_prev.so, alternating, bench.py each time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  for lib in "" "$R/dd3d_amd/lib/libdd3d_hip_prev.so"; do
    DD3D_HIP_LIB=$lib python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-new }'[-8:], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
