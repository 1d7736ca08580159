"""Dev tool: time one op of the DD3D-DLA34 plan (default: towers.1) on the data the forward left in its buffers and again on
all-zero operands (same instruction stream, minimal switching power: the gap is what the power-limited shader clock costs).

    DD3D_MATH=bf16x3 python tests/gpu_tower_probe.py [op names, comma separated]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def time_op(plan, op, iters=40):
    st = hip.current_stream()
    for _ in range(10):
        op(plan.lib, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            op(plan.lib, st)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["towers.1"]
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(1, 384, 1280))
    plan.run()
    torch.cuda.synchronize()
    ops = [op for op in plan.ops if op.name in names]
    real = {op.name: time_op(plan, op) for op in ops}
    for b in plan.bufs.values():
        if b.t is not None:
            b.t.zero_()
        if b.p is not None:
            b.p.zero_()
    for entry in plan._split.values():
        entry[1].zero_()
    zero = {op.name: time_op(plan, op) for op in ops}
    for op in ops:
        fl = 2.0 * op.macs
        print(f"{os.environ.get('DD3D_MATH', 'bf16x3'):7s} planes={int(plan.use_planes)} {op.name:24s} data {real[op.name]:8.2f} us ({fl / real[op.name] / 1e6:6.1f} TF)   "
              f"zeros {zero[op.name]:8.2f} us ({fl / zero[op.name] / 1e6:6.1f} TF)   ratio {real[op.name] / zero[op.name]:.3f}")


if __name__ == "__main__":
    main()
