"""Per-op timing of the DD3D-DLA34 launch plan on the GPU (HIP events on the launch stream).

    python tests/gpu_conv_bench.py [H W B]    -> table + gpurun_out/conv_bench[_TAG].json   (TAG = $DD3D_BENCH_TAG)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.engine import ConvOp  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def time_op(plan, op, iters=30):
    st = hip.current_stream()
    for _ in range(3):
        op(plan.lib, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op(plan.lib, st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, H, W))
    plan.run()
    torch.cuda.synchronize()
    rows, tot_us, tot_flop = [], 0.0, 0.0
    for op in plan.ops:
        us = time_op(plan, op)
        flop = 2.0 * op.macs
        info = op.info if isinstance(op, ConvOp) else {"name": op.name}
        rows.append(dict(info, us=round(us, 2), tflops=round(flop / us / 1e6, 2) if flop else 0.0))
        tot_us += us
        tot_flop += flop
    for r in rows:
        print(f"{r['name']:28s} M={r.get('M', 0):7d} N={r.get('N', 0):4d} K={r.get('K', 0):5d} tile={str(r.get('tile', '')):10s} sk={r.get('splitk', 0):2d} "
              f"blocks={r.get('blocks', 0):5d} {r['us']:9.2f} us {r['tflops']:7.2f} TF/s")
    print(f"sum of op times {tot_us / 1e3:.3f} ms; {tot_flop / tot_us / 1e6:.2f} TFLOP/s over the op sum")
    plan.capture()
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        plan.run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print(f"graph replay {ms:.3f} ms / forward = {B / ms * 1e3:.1f} img/s")
    tag = os.environ.get("DD3D_BENCH_TAG", "")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/conv_bench{('_' + tag) if tag else ''}.json", "w") as f:
        json.dump({"rows": rows, "sum_ms": tot_us / 1e3, "graph_ms": ms}, f, indent=1)


if __name__ == "__main__":
    main()
