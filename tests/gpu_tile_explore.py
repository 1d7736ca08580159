"""Dev tool: time every (tile, split-K) candidate for each distinct conv of the DD3D-DLA34 plan on the GPU.

    python tests/gpu_tile_explore.py [H W B] > gpurun_out/tile_explore.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd import engine  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402

RECORD = []
_orig_init = engine.ConvOp.__init__


def _rec_init(self, plan, meta, stride, pad, segs, relu, tile=None, splitk=None, name="", **kw):
    _orig_init(self, plan, meta, stride, pad, segs, relu, tile=tile, splitk=splitk, name=name, **kw)
    RECORD.append((name, plan, meta, stride, pad, segs, relu, kw))


def time_op(plan, op, iters=30):
    st = hip.current_stream()
    best = float("inf")
    for _ in range(2):
        for _ in range(5):
            op(plan.lib, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            op(plan.lib, st)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    global exp
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    exp = os.environ.get("DD3D_EXP", "dd3d_kitti_dla34")
    cfg = get_cfg(exp)
    model = build_model(cfg)
    tag = {"dd3d_kitti_dla34": "dla34_kitti", "dd3d_kitti_v99": "v99_kitti", "dd3d_nusc_dla34": "dla34_nusc", "dd3d_nusc_v99": "v99_nusc"}.get(exp, "dla34_kitti")
    model.load_state_dict(make_state_dict(model, calib=load_calib(tag)))
    model.use_graph = False
    engine.ConvOp.__init__ = _rec_init
    plan, _ = model.stage_inputs(make_inputs(B, H, W, dataset="nusc" if "nusc" in exp else "kitti"))
    engine.ConvOp.__init__ = _orig_init
    plan.run()
    torch.cuda.synchronize()
    seen = {}
    table = {}
    for t in list(engine.TILE_TABLE.values()) + list(engine.PLANE_TILE_TABLE.values()):
        t.clear()  # measure against the analytic model (in place: dd3d_amd.engine.tiling holds the same dict objects)
    only = os.environ.get("DD3D_EXPLORE_ONLY", "")  # comma-separated op-name prefixes (e.g. the big launches whose other tiles the block limit skips)
    max_blocks = int(os.environ.get("DD3D_EXPLORE_MAXBLOCKS", "6000"))
    for name, pl, meta, stride, pad, segs, relu, kw in RECORD:
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        m_list = tuple(s["out"].B * s["out"].H * s["out"].W for s in segs)
        key = (m_list, meta["N"], meta["Kpad"], meta["Cin"], stride)
        if key in seen:
            continue
        seen[key] = name
        nk = meta["Kpad"] // 32
        planes = pl.use_planes and meta["Cin"] % 32 == 0 and all(sg["in"].np for sg in segs)
        math = pl.math if (meta["Cin"] % 32 == 0 and (meta["N"] > 32 or planes)) else hip.MATH_F32
        cur_cfg, cur_sk = engine.choose_tiling(list(m_list), meta["N"], meta["Kpad"], stride, math, planes=planes)
        res = []
        for cfg_id in (engine.PLANE_TILES if planes else engine.MATH_TILES[math]):
            bm, bn = hip.TILE_SHAPES[cfg_id]
            if (bn == 32) != (meta["N"] <= 32) and not (planes and bn == 64 and meta["N"] <= 32):
                continue
            if bn == 128 and meta["N"] <= 64:
                continue
            blocks = sum(-(-m // bm) for m in m_list) * -(-meta["N"] // bn)
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if sk > 1 and (nk // sk < 3 or blocks * sk > 1300):
                    continue
                if sk == 1 and blocks > max_blocks and cfg_id != cur_cfg:
                    continue
                n_amax = len(pl.amax_names)
                try:
                    op = engine.ConvOp(pl, meta, stride, pad, segs, relu, tile=cfg_id, splitk=sk, name=name, math=math, **{k: v for k, v in kw.items() if k != 'math'})
                    us = time_op(pl, op)
                    del pl.amax_names[n_amax:]  # (every candidate registers a range-guard slot: 512 per plan; ten images per launch ran out of them)
                except Exception as e:  # noqa: BLE001
                    us = float("nan")
                    print(f"  !! {name} tile={hip.TILE_NAMES[cfg_id]} splitk={sk}: {type(e).__name__}: {str(e)[:300]}", flush=True)
                res.append((us, hip.TILE_NAMES[cfg_id], cfg_id, sk, blocks * sk))
        res.sort()
        cur = [r for r in res if r[2] == cur_cfg and r[3] == cur_sk]
        cur_us = cur[0][0] if cur else float("nan")
        best = res[0]
        if cur_us > 1.03 * best[0]:  # only keep entries that beat the model by more than the measurement noise
            table[engine.tile_key(m_list, meta["N"], meta["Kpad"], stride)] = [best[1], best[3], round(best[0], 1), round(cur_us, 1)]
        print(f"{name:26s} M={sum(m_list):7d} N={meta['N']:4d} K={meta['Kpad']:5d} s{stride} model=({hip.TILE_NAMES[cur_cfg]},sk{cur_sk}) {cur_us:7.2f}us "
              f"best=({best[1]},sk{best[3]},{best[4]}blk) {best[0]:7.2f}us gain {cur_us - best[0]:6.2f} | "
              + " ".join(f"{r[1]}/{r[3]}:{r[0]:.1f}" for r in res[:7]), flush=True)


    import json
    mname = next(k for k, v in engine.MATH_NAMES.items() if v == plan.math)
    out = os.path.join("gpurun_out", f"tile_table_{exp}_{H}x{W}_b{B}_{mname}{'_planes' if plan.use_planes else ''}{'_only' if only else ''}.json")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("wrote", out, len(table), "entries")


if __name__ == "__main__":
    main()
