"""Stage-by-stage parity report of the HIP forward against the oracle (run on the GPU box):

    python tests/gpu_stage_check.py [H W [B]]      -> prints one line per stage and writes gpurun_out/stage_check.json
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402
from oracle import dd3d_oracle as O  # noqa: E402
from tests.util import candidates_from_plan, max_abs, oracle_heads_to_plan, quat_err, rel_err  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    report = {"H": H, "W": W, "B": B}
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_kitti"))
    model.load_state_dict(sd)
    model.use_graph = False
    inputs = make_inputs(B, H, W)
    t0 = time.time()
    ref, st = O.dd3d_forward(sd, cfg, inputs)
    report["oracle_s"] = time.time() - t0
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()

    def line(name, a, b):
        e, r = max_abs(a, b), rel_err(a, b)
        report[name] = {"max_abs": e, "rel": r, "ref_absmax": float(b.abs().max())}
        print(f"{name:28s} max_abs={e:.3e} rel={r:.3e} (|ref|max={float(b.abs().max()):.3e})", flush=True)

    line("images", plan.normalized_image(), st["images"])
    for k, v in st["bottom_up"].items():
        line("bottom_up." + k, plan.bottom_up[k].nchw(), v)
    for l, f in enumerate(st["features"]):
        line(f"fpn.{l}", plan.features[l].nchw(), f)
    C = cfg.DD3D.NUM_CLASSES
    for l in range(len(st["features"])):
        line(f"logits.{l}", plan.cls_maps[l].nchw(0, C), st["logits"][l])
        line(f"box2d_reg.{l}", plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l])
        line(f"centerness.{l}", plan.b2d_maps[l].nchw(4, 1), st["centerness"][l])
        fused = torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1)
        line(f"box3d.{l}", plan.b3d_maps[l].t[..., :11 * C].permute(0, 3, 1, 2), fused)
    report["npass_hip"] = plan.npass.cpu().tolist()
    report["npass_ref"] = [[len(info[i]["fg_inds"]) for info in st["level_info"]] for i in range(B)]
    print("npass hip", report["npass_hip"], "ref", report["npass_ref"])
    out = model.collect(plan, inputs, image_sizes)
    report["ndet_hip"] = [len(o["instances"]) for o in out]
    report["ndet_ref"] = [len(r["scores"]) for r in ref]
    print("detections hip", report["ndet_hip"], "ref", report["ndet_ref"])

    # ---- integer parity on identical head maps: feed the oracle's head maps to the HIP post-processing
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=len(plan.ops) - 2)
    torch.cuda.synchronize()
    out2 = model.collect(plan, inputs, image_sizes)
    for i in range(B):
        c = candidates_from_plan(plan, i)
        rc = st["candidates"][i]
        same_n = len(c["scores"]) == len(rc["scores"])
        print(f"[img {i}] candidates hip={len(c['scores'])} ref={len(rc['scores'])}")
        if same_n and len(rc["scores"]):
            report[f"cand{i}"] = {
                "classes_equal": bool(torch.equal(c["pred_classes"], rc["pred_classes"])),
                "levels_equal": bool(torch.equal(c["fpn_levels"], rc["fpn_levels"])),
                "box": max_abs(c["pred_boxes"], rc["pred_boxes"]), "score": max_abs(c["scores"], rc["scores"]),
                "score3d": max_abs(c["scores_3d"], rc["scores_3d"]), "quat": quat_err(c["quat"], rc["pred_boxes3d"]["quat"]),
                "ctr": max_abs(c["proj_ctr"], rc["pred_boxes3d"]["proj_ctr"]), "depth": rel_err(c["depth"], rc["pred_boxes3d"]["depth"]),
                "size": rel_err(c["size"], rc["pred_boxes3d"]["size"])
            }
            print("   ", report[f"cand{i}"])
        o, r = out2[i]["instances"], ref[i]
        print(f"[img {i}] final hip={len(o)} ref={len(r['scores'])}")
        if len(o) == len(r["scores"]) and len(o):
            report[f"final{i}"] = {
                "classes_equal": bool(torch.equal(o.pred_classes.cpu(), r["pred_classes"])),
                "levels_equal": bool(torch.equal(o.fpn_levels.cpu(), r["fpn_levels"])),
                "box": max_abs(o.pred_boxes.tensor, r["pred_boxes"]), "score3d": max_abs(o.scores_3d, r["scores_3d"]),
                "quat": quat_err(o.pred_boxes3d.quat, r["pred_boxes3d"]["quat"]),
                "depth": rel_err(o.pred_boxes3d.depth, r["pred_boxes3d"]["depth"])
            }
            print("   ", report[f"final{i}"])

    # ---- timing: eager launches vs one hipGraph replay
    plan2, _ = model.stage_inputs(inputs)
    for mode in ("eager", "graph"):
        if mode == "graph":
            plan2.capture()
        for _ in range(3):
            plan2.run()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 20
        for _ in range(n):
            plan2.run()
        torch.cuda.synchronize()
        ms = (time.time() - t0) / n * 1e3
        report[f"ms_{mode}"] = ms
        print(f"{mode}: {ms:.3f} ms / forward  ({B / ms * 1e3:.1f} img/s)")
    report["convs"] = plan2.describe()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/stage_check.json", "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
