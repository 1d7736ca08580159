#!/bin/bash
# Build the library of a git revision (default HEAD) next to the working-tree one: dd3d_amd/lib/libdd3d_hip_prev.so
# (same ABI assumed), for A/B runs on one GPU box:  DD3D_HIP_LIB=dd3d_amd/lib/libdd3d_hip_prev.so python bench.py
REV=${1:-HEAD}
T=$(mktemp -d)
git archive $REV dd3d_amd/csrc include | tar -x -C $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$T/include -I$T/dd3d_amd/csrc $T/dd3d_amd/csrc/*.hip -o dd3d_amd/lib/libdd3d_hip_prev.so && echo built prev from $REV
rm -rf $T
