"""Launch a few selected ops of the DD3D-DLA34 plan repeatedly (for rocprofv3 --pmc passes).

    python tests/gpu_pmc_probe.py towers.1,level3.tree1.tree1.conv2 [iters] [images per launch plan]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["towers.1"]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, 384, 1280))
    plan.run()  # fill every buffer with realistic data
    torch.cuda.synchronize()
    st = hip.current_stream()
    for op in plan.ops:
        if op.name in names:
            for _ in range(iters):
                op(plan.lib, st)
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
