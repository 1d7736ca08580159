"""Dev tool: duration of selected launches of the DD3D-DLA34 plan (HIP events around a burst of back-to-back launches on the launch stream).

    [DD3D_TIME_LIB=build/ab/libdd3d_<variant>.so] python tests/gpu_op_time.py [H W B] [name-prefix ...]

DD3D_TIME_LIB: the forward that fills the buffers runs on the shipped library, the TIMED launches on the variant -- so an ablated kernel
(wrong results) is timed on the real operand data: the matrix pipe's rate depends on the operand bits (zeros run 1.3 - 1.7x faster than
realistic activations), and a variant library that also produced the inputs would time its K loop on zeros or garbage.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

if not os.environ.get("DD3D_HIP_LIB"):
    g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.engine import ConvOp  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402

DEFAULT = ["level2.tree1.conv2", "level2.root", "level3.tree1.tree1.conv2", "level3.tree2.root", "level4.tree1.tree1.conv2", "level5.tree1.conv2",
           "fpn_lateral3", "fpn_output3", "towers.0", "predictors"]


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    want = sys.argv[4:] or DEFAULT
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, H, W))
    plan.run()
    torch.cuda.synchronize()
    st = hip.current_stream()
    lib = plan.lib
    if os.environ.get("DD3D_TIME_LIB"):
        import ctypes
        lib = ctypes.CDLL(os.environ["DD3D_TIME_LIB"])
    for op in plan.ops:
        if not any(op.name.startswith(w) for w in want):
            continue
        for _ in range(5):
            op(lib, st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                op(lib, st)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        info = op.info if isinstance(op, ConvOp) else {}
        print(f"{op.name:28s} {best:8.2f} us   tile {info.get('tile_name', '')} sk {info.get('splitk', '')} blocks {info.get('blocks', '')}", flush=True)


if __name__ == "__main__":
    main()
