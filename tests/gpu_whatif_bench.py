"""Dev tool (GPU): what the shipped issue mode's throughput WOULD be if parts of the forward cost nothing -- bench.py with the launches whose names
start with one of DD3D_WHATIF_SKIP's prefixes replaced by no-ops (their outputs stay zero: results are garbage, only the clock is read).

Answers "which part of the forward bounds the driver command?" in situ -- with five slots in flight a launch's isolated duration says little about
what removing it would return.  An upper bound on what ANY optimisation of those launches can buy.

    DD3D_WHATIF_SKIP=level3.,level4.,level5. python tests/gpu_whatif_bench.py --steps 20 --warmup 5
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets GPU_MAX_HW_QUEUES before the runtime starts)
from dd3d_amd.engine import ops as O  # noqa: E402

SKIP = tuple(p for p in os.environ.get("DD3D_WHATIF_SKIP", "").split(",") if p)
_append = O.OpList.append


def append(self, op):
    if SKIP and op.name.startswith(SKIP):
        op.__class__ = type("Skipped" + op.__class__.__name__, (op.__class__, ), {"__call__": lambda self, lib, stream: None})
    _append(self, op)


O.OpList.append = append
bench.verify_work = lambda runner, plan, B: {"skipped": list(SKIP)}  # (garbage results: nothing to verify)

if __name__ == "__main__":
    sys.argv += ["--no-cpu-baseline"]
    bench.main()
