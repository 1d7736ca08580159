#!/bin/bash
# Dev tool: registers / spills / occupancy of every kernel of one translation unit, as the compiler reports them.
#   bash tests/tools/kernel_resources.sh dd3d_amd/csrc/conv_planes_row.hip [-D...]
R=$(cd "$(dirname "$0")/../.." && pwd)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/dd3d_amd/csrc "$@" -c $src -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import sys, re, subprocess
rows, cur = [], None
for ln in sys.stdin:
    if 'error' in ln: print(ln.rstrip())
    m = re.search(r'remark:\s+(.*?) \[-Rpass', ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':',1); cur[k.strip()] = v.strip()
for r in rows:
    nm = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    nm = re.sub(r'\(dd3d::ConvKArgs\)|void |dd3d::', '', nm)
    print(f\"{nm:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} sgpr-spill {r.get('SGPRs Spill','?'):>3s} occ {r.get('Occupancy [waves/SIMD]','?')}\")
"
