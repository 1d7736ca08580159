"""Dev tool: what every arithmetic mode of the convolutions costs and delivers, measured -- the table DESIGN.md section 6 quotes.

For each mode (f32 MFMA, bf16x3, f16x2, bf16x2, bf16) and each model (DD3D-DLA34 KITTI, DD3D-V2-99 KITTI at 384x1280 B=1):
  * one 3x3 256->256 convolution against a float64 convolution of the same f32 data (max |err| / max |ref|),
  * the full forward against the CPU oracle (the reference restated in fp32): max abs error of every head map relative to its
    largest entry, candidate-membership flips and how far from the cut they sit, relative errors of the decoded 2D box / depth /
    size / 8-corner L1 on the detections both sides produce,
  * time per forward (hipGraph replay, one step at a time).

    python tests/gpu_math_modes.py [dla34|v99|both] [H W]      -> gpurun_out/math_modes.json + a printed table
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg, hip  # noqa: E402
from dd3d_amd.engine import MATH_NAMES, ConvOp, PlanBase, pack_filter  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402
from oracle import dd3d_oracle as O  # noqa: E402  (test infrastructure: the checker, not the thing measured)
from tests.util import candidate_margins, max_abs, rel_err  # noqa: E402

MODES = ["f32", "bf16x3", "f16x2", "bf16x2", "bf16"]


def conv_vs_float64():
    gen = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 1, 24, 40, 256, 256
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / 48.0
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    out = {}
    for name in MODES:
        plan = PlanBase("cuda")
        plan.math = MATH_NAMES[name]
        wp, meta = pack_filter(w, plan.device)
        xin, yout = plan.buf("x", B, H, W, Cin, kind="both"), plan.buf("y", B, H, W, Cout)
        xin.t.copy_(x.permute(0, 2, 3, 1))
        if plan.use_planes:
            plan.split(xin.view(), name="x.split")
        ones, zeros = torch.ones(Cout, device=plan.device), torch.zeros(Cout, device=plan.device)
        plan.ops.append(ConvOp(plan, meta, 1, 1, [{"in": xin.view(), "out": yout.view(), "w": wp, "scale": ones, "bias": zeros}], False, name="acc"))
        plan.launch()
        torch.cuda.synchronize()
        d = (yout.nchw().cpu().double() - ref).abs()
        out[name] = dict(max_err_over_max_ref=float(d.max() / ref.abs().max()), rms_err_over_rms_ref=float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
    return out


def forward_study(exp, tag, H, W):
    cfg = get_cfg(exp)
    base = build_model(cfg)
    sd = make_state_dict(base, calib=load_calib(tag))
    inputs = make_inputs(1, H, W)
    t0 = time.time()
    with torch.no_grad():
        ref, st = O.dd3d_forward(sd, cfg, inputs)
    print(f"[{exp}] oracle forward {time.time() - t0:.1f} s, {len(ref[0]['scores'])} detections", flush=True)
    C = cfg.DD3D.NUM_CLASSES
    rows = {}
    for name in MODES:
        model = build_model(cfg)
        model.load_state_dict(sd)
        model.math = name
        model.use_graph = True
        out = model(inputs)[0]["instances"]
        plan = next(iter(model._plans.values()))
        torch.cuda.synchronize()
        r = {}
        # head maps: max abs error relative to the map's largest entry, worst over the levels
        worst = {}
        for l in range(len(st["logits"])):
            for key, got, want in [("logits", plan.cls_maps[l].nchw(0, C), st["logits"][l]), ("box2d_reg", plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l]),
                                   ("centerness", plan.b2d_maps[l].nchw(4, 1), st["centerness"][l]),
                                   ("depth", plan.b3d_maps[l].nchw(6 * C, C), st["depth"][l]), ("quat", plan.b3d_maps[l].nchw(0, 4 * C), st["quat"][l]),
                                   ("size", plan.b3d_maps[l].nchw(7 * C, 3 * C), st["size"][l])]:
                e = max_abs(got, want) / max(1.0, float(want.abs().max()))
                worst[key] = max(worst.get(key, 0.0), e)
        r["head_map_err"] = worst
        n_hip, n_ref, margins = candidate_margins(plan, st, cfg, 0)
        r["candidates"] = dict(hip=n_hip, oracle=n_ref, flips=len(margins), max_distance_from_cut=max(margins, default=0.0))
        # detections both sides produce
        key = lambda lv, loc, cl: [(int(a), float(x), float(y), int(c)) for a, (x, y), c in zip(lv.tolist(), loc.tolist(), cl.tolist())]
        ko = key(out.fpn_levels.cpu(), out.locations.cpu(), out.pred_classes.cpu())
        kr = key(ref[0]["fpn_levels"], ref[0]["locations"], ref[0]["pred_classes"])
        common = [k for k in ko if k in set(kr)]
        io, ir = [ko.index(k) for k in common], [kr.index(k) for k in common]
        b = ref[0]["pred_boxes3d"]
        r["detections"] = dict(hip=len(ko), oracle=len(kr), common=len(common))
        if common:
            tv = O.boxes3d_tvec(b)
            c_ref = O.boxes3d_corners(b["quat"], tv, b["size"])[ir]
            c_got = out.pred_boxes3d.to("cpu").corners[io]
            r["decoded_rel_err"] = dict(
                box2d=max_abs(out.pred_boxes.tensor[io], ref[0]["pred_boxes"][ir]) / max(1.0, float(ref[0]["pred_boxes"].abs().max())),
                depth=rel_err(out.pred_boxes3d.depth[io], b["depth"][ir]), size=rel_err(out.pred_boxes3d.size[io], b["size"][ir]),
                score_3d=rel_err(out.scores_3d[io], ref[0]["scores_3d"][ir]),
                corners_l1=float((c_got - c_ref).abs().mean() / max(1.0, float(c_ref.abs().mean()))))
        # time
        for _ in range(5):
            plan.run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.run()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        r["ms_per_forward"] = best
        r["tflops_f32_equiv"] = 2 * plan.conv_macs / best / 1e9
        rows[name] = r
        d = r.get("decoded_rel_err", {})
        print(f"[{exp}] {name:7s} {best:7.3f} ms  head maps max err/max: " + " ".join(f"{k}={v:.1e}" for k, v in worst.items()) +
              f" | candidates {n_hip}/{n_ref} flips {len(margins)} (max dist {max(margins, default=0.0):.1e}) | dets {len(ko)}/{len(kr)} common {len(common)} | "
              + " ".join(f"{k}={v:.1e}" for k, v in d.items()), flush=True)
        del model, plan
        torch.cuda.empty_cache()
    return rows


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
    res = {"conv3x3_256_vs_float64": conv_vs_float64()}
    for k, v in res["conv3x3_256_vs_float64"].items():
        print(f"conv vs float64  {k:7s} max err / max ref {v['max_err_over_max_ref']:.2e}   rms err / rms ref {v['rms_err_over_rms_ref']:.2e}", flush=True)
    if which in ("dla34", "both"):
        res["dd3d_kitti_dla34"] = forward_study("dd3d_kitti_dla34", "dla34_kitti", H, W)
    if which in ("v99", "both"):
        res["dd3d_kitti_v99"] = forward_study("dd3d_kitti_v99", "v99_kitti", H, W)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/math_modes.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
