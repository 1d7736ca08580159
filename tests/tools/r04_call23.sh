#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
for v in "" stem_16_4 stem_16_8 stem_32_4; do
  echo "== ${v:-shipped (8 x 32 tile, 8 waves)}" | tee -a $O/stem_variants.txt
  DD3D_HIP_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 300 python -m pytest tests/test_stem_fused_gpu.py -q -m gpu -x 2>&1 | tail -1 | tee -a $O/stem_variants.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 4 stem 2>&1 | grep " us " | tee -a $O/stem_variants.txt
  DD3D_TIME_LIB=${v:+$R/build/ab/libdd3d_$v.so} timeout 200 python tests/gpu_op_time.py 384 1280 1 stem 2>&1 | grep " us " | tee -a $O/stem_variants.txt
done
