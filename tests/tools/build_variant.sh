#!/bin/bash
# build/ab/libdd3d_<name>.so = the working-tree library with extra -D knobs (csrc/build_flags.h lists them; every knob computes correct
# results) -- for A/B measurements on one box: DD3D_HIP_LIB=build/ab/libdd3d_<name>.so selects it (dd3d_amd/hip.py refuses a library that
# reports knobs unless it is chosen that way).
#   usage: build_variant.sh <name> [--ablations] [-DDD3D_...]...
# --ablations: build from a COPY of csrc with tests/tools/variants/r04_timing_ablations.patch applied: the round-4 timing experiments that
# remove one ingredient of the K loop / epilogue (-DDD3D_ABLATE_DMA, _MFMA, _DSREAD, _DSREAD_LOOP, _EPI, _EPI_VALU, _EPI_STORE,
# -DDD3D_ROW_NOMASK, -DDD3D_EXP_GROUP_BARRIER).  Those compute WRONG results and exist for timing only; the product sources cannot build them.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
SRC=$R/dd3d_amd/csrc
mkdir -p $R/build/ab
if [ "$1" == "--ablations" ]; then
  shift
  rm -rf $R/build/ab/src_$name && mkdir -p $R/build/ab/src_$name/dd3d_amd && cp -r $SRC $R/build/ab/src_$name/dd3d_amd/csrc
  (cd $R/build/ab/src_$name && patch -s -p1 < $R/tests/tools/variants/r04_timing_ablations.patch)
  SRC=$R/build/ab/src_$name/dd3d_amd/csrc
fi
objs=""
for f in $SRC/*.hip; do
  o=$R/build/ab/$(basename ${f%.hip})_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$SRC "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/build/ab/libdd3d_$name.so
echo "built $R/build/ab/libdd3d_$name.so"
