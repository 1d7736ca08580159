"""Dev tool (round 6): where a block of the row-shared 3x3 kernel spends its time -- s_memtime stamps of every wave's phases, from a variant
library built with -DDD3D_ROW_STAMP=1 (results unchanged; the stamps cost ~10 % of the wave cycles):

    bash tests/tools/build_variant.sh stamp -DDD3D_ROW_STAMP=1
    DD3D_HIP_LIB=build/ab/libdd3d_stamp.so python tests/gpu_row_stamp_probe.py [B] [op name substring ...]

For each chosen convolution of the B-image DLA-34 384 x 1280 plan (launched alone, after the whole plan ran once): per wave, averaged over the
launch's first 512 blocks, in shader cycles (s_memtime ticks); `launch` in us (HIP events, with the stamps' overhead)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.engine import ConvOp  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    picks = sys.argv[2:] or ["level2.tree1.conv2", "level3.tree1.tree1.conv2", "level4.tree1.tree1.conv2", "level5.tree1.conv2", "fpn_outputs", "towers.1",
                             "predictors"]
    lib = hip.lib()
    fn = lib.dd3d_debug_row_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, 384, 1280))
    plan.run()
    torch.cuda.synchronize()
    buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    # tick length: time a known launch with events against its stamps
    print(f"# {B} image(s); tile policy {plan.tile_policy}")
    print(f"{'op':28s} {'tile':>10s} {'blocks':>6s} {'launch':>8s} | {'block':>7s} = {'prologue':>8s} + {'1st data':>8s} + {'K loop':>7s} + {'epilogue':>8s} | per step: {'own work':>8s} {'vmcnt':>6s} {'barrier':>7s}  steps")
    for k, op in enumerate(plan.ops):
        if not isinstance(op, ConvOp) or not any(p in op.name for p in picks):
            continue
        for _ in range(2):
            plan.launch(k, k + 1)
        torch.cuda.synchronize()
        assert fn(buf.ctypes.data, buf.size) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.launch(k, k + 1)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        assert fn(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(512, 8, 8).astype(np.float64)
        ok = st[:, :, 4] > 0
        if not ok.any():
            print(f"{op.name:28s} (no stamps: not the row kernel)")
            continue
        w = st[ok]
        span = (st[:, :, 4][ok].max() - st[:, :, 0][ok].min())
        tick = us / span if span > 0 else 0.0  # us per tick, from the launch's own span (blocks beyond 512 unrecorded: an upper bound)
        blk = (w[:, 4] - w[:, 0]).mean()
        pro = (w[:, 1] - w[:, 0]).mean()
        first = (w[:, 2] - w[:, 1]).mean()
        loop = (w[:, 3] - w[:, 2]).mean()
        epi = (w[:, 4] - w[:, 3]).mean()
        nk = op.info["K"] // 32
        sk = max(int(op.info.get("splitk", 1)), 1)
        steps = max(nk // sk, 1)
        f = 1.0  # s_memtime ticks = shader cycles (MI355X_MICROARCH.md)
        print(f"{op.name:28s} {op.info['tile_name']:>10s} {op.info.get('blocks', 0):6d} {us:8.1f} | {blk:7.0f} = {pro:8.0f} + {first:8.0f} + {loop:7.0f} + {epi:8.0f} | "
              f"{w[:, 5].mean() / steps:8.0f} {w[:, 6].mean() / steps:6.0f} {w[:, 7].mean() / steps:7.0f} cyc {steps}")
        per_wave = []
        for wv in range(8):
            m = ok[:, wv]
            if m.any():
                x = st[:, wv][m]
                per_wave.append(f"w{wv}: {x[:, 5].mean() / steps:.0f}/{x[:, 6].mean() / steps:.0f}/{x[:, 7].mean() / steps:.0f}")
        print("      per wave own work / vmcnt / barrier per step:  " + "  ".join(per_wave))


if __name__ == "__main__":
    main()
