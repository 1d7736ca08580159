"""Dev tool: true incremental cost of every op INSIDE a hipGraph: capture the launch-plan prefixes ops[:k] for all k, replay
each, and print the differences (isolated per-op timings overstate the single-block post-processing kernels and miss
inter-kernel gaps).

    python tests/gpu_prefix_bench.py [H W B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    exp = os.environ.get("DD3D_EXP", "dd3d_kitti_dla34")
    cfg = get_cfg(exp)
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti" if "dla34" in exp else "v99_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, H, W))
    plan.run()
    torch.cuda.synchronize()
    times = []
    side = torch.cuda.Stream()
    for k in range(1, len(plan.ops) + 1):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                plan.launch(0, k)
        best = 1e9
        for _ in range(3):
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                gr.replay()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        times.append(best)
    prev = 0.0
    for op, t in zip(plan.ops, times):
        print(f"{op.name:28s} +{t - prev:8.2f} us   (prefix {t:9.2f} us)")
        prev = t


if __name__ == "__main__":
    main()
