#!/bin/bash
# build/ab/libdd3d_<name>.so = the working-tree library with extra -D flags on csrc/conv_planes.hip (A/B measurements on one box:
# DD3D_HIP_LIB=build/ab/libdd3d_<name>.so selects it).  usage: build_variant.sh <name> [-DDD3D_...]...
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/dd3d_amd/csrc "$@" -c $R/dd3d_amd/csrc/conv_planes.hip -o $R/build/ab/cp_$name.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/dd3d_amd/csrc "$@" -c $R/dd3d_amd/csrc/conv_planes_row.hip -o $R/build/ab/cpr_$name.o &
wait
objs=$(ls $R/build/obj/*.o | grep -v conv_planes.o | grep -v conv_planes_row.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build/ab/cp_$name.o $R/build/ab/cpr_$name.o -o $R/build/ab/libdd3d_$name.so
