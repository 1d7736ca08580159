"""Post-head configuration switches that the reference's experiments leave at their defaults: THRESH_WITH_CTR False
(fcos2d.py:296-300), FEATURE_LOCATIONS_OFFSET "half" (tensor2d.py:20-24), PREDICT_DISTANCE True and SCALE_DEPTH_BY_FOCAL_LENGTHS
False (fcos3d.py:32-40), PREDICT_ALLOCENTRIC_ROT False (fcos3d.py:42-46), CLASS_AGNOSTIC_BOX3D True (fcos3d.py:385-390).
Goldens: the reference's own DD3D with those switches (tests/golden/make_golden.py variants).  (Named to run last.)"""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_golden import VARIANTS, case_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HEADS = ("logits", "box2d_reg", "centerness", "quat", "ctr", "depth", "size", "conf")


def _load(name):
    g = np.load(os.path.join(GOLD, f"dla34_kitti_variant_{name}.npz"))
    return g, (lambda k: torch.from_numpy(g[k]))


@pytest.mark.parametrize("name", list(VARIANTS))
def test_oracle_variant_matches_reference_golden(name):
    from oracle import dd3d_oracle as O
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", VARIANTS[name])
    g, t = _load(name)
    with torch.no_grad():
        res, st = O.dd3d_forward(sd, cfg, case_inputs(1, 128, 256, False, "kitti"))
    for l in range(5):
        for k in HEADS:
            assert st[k][l].shape == t(f"{k}{l}").shape and torch.allclose(st[k][l], t(f"{k}{l}"), rtol=1e-5, atol=2e-5), (k, l)
    r = res[0]
    assert len(r["scores"]) == len(g["det0_scores"]) > 0
    assert torch.equal(r["pred_classes"], t("det0_classes")) and torch.equal(r["fpn_levels"], t("det0_levels"))
    assert torch.equal(r["locations"], t("det0_locations"))
    assert torch.allclose(r["pred_boxes"], t("det0_boxes"), rtol=1e-5, atol=1e-4)
    assert torch.allclose(r["scores"], t("det0_scores"), rtol=1e-5) and torch.allclose(r["scores_3d"], t("det0_scores_3d"), rtol=1e-5)
    b = r["pred_boxes3d"]
    assert torch.allclose(b["quat"], t("det0_quat"), atol=1e-5) and torch.allclose(b["proj_ctr"], t("det0_proj_ctr"), rtol=1e-5, atol=1e-4)
    assert torch.allclose(b["depth"], t("det0_depth"), rtol=1e-5) and torch.allclose(b["size"], t("det0_size"), rtol=1e-5)
    assert torch.allclose(O.boxes3d_tvec(b), t("det0_tvec"), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("name", list(VARIANTS))
def test_hip_variant_matches_reference_golden(hiplib, name):
    from tests.util import bundle, gpu_model, max_abs, oracle_heads_to_plan, quat_err, rel_err
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", VARIANTS[name])
    g, t = _load(name)
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = case_inputs(1, 128, 256, False, "kitti")
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    C3 = 1 if cfg.DD3D.FCOS3D.CLASS_AGNOSTIC_BOX3D else C
    st = {k: [t(f"{k}{l}") for l in range(5)] for k in HEADS}
    for l in range(5):
        assert max_abs(plan.cls_maps[l].nchw(0, C), st["logits"][l]) < 1e-4 * max(1.0, float(st["logits"][l].abs().max()))
        assert max_abs(plan.b3d_maps[l].nchw(6 * C3, C3), st["depth"][l]) < 1e-4 * max(1.0, float(st["depth"][l].abs().max()))
    oracle_heads_to_plan(plan, st, C)  # integer parity on identical head maps
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    o = model.collect(plan, inputs, image_sizes)[0]["instances"]
    assert torch.equal(o.pred_classes.cpu(), t("det0_classes")) and torch.equal(o.fpn_levels.cpu(), t("det0_levels"))
    assert torch.equal(o.locations.cpu(), t("det0_locations"))
    assert max_abs(o.pred_boxes.tensor, t("det0_boxes")) < 1e-3 * max(1.0, float(t("det0_boxes").abs().max()))
    assert rel_err(o.scores_3d, t("det0_scores_3d")) < 1e-3 and rel_err(o.pred_boxes3d.depth, t("det0_depth")) < 1e-3
    assert rel_err(o.pred_boxes3d.size, t("det0_size")) < 1e-3 and quat_err(o.pred_boxes3d.quat, t("det0_quat")) < 1e-3
    assert max_abs(o.pred_boxes3d.tvec, t("det0_tvec")) < 1e-3 * max(1.0, float(t("det0_tvec").abs().max()))


@pytest.mark.parametrize("name", list(VARIANTS))
def test_plan_of_the_variant_builds(hiplib, name):
    """Host side (no GPU): the model takes the reference-shaped state dict strictly and its launch plan builds."""
    from dd3d_amd import META_ARCH_REGISTRY
    from dd3d_amd.engine import ForwardPlan
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", VARIANTS[name])
    model = META_ARCH_REGISTRY.get("DD3D")(cfg)
    model.load_state_dict(sd, strict=True)
    plan = ForwardPlan(model, 1, 128, 256, device="cpu", dry_run=True)
    C = cfg.DD3D.NUM_CLASSES
    C3 = 1 if cfg.DD3D.FCOS3D.CLASS_AGNOSTIC_BOX3D else C
    assert plan.b3d_pitch >= 11 * C3 and plan.cls_pitch >= C
    a = plan.select_args
    assert a.thresh_with_ctr == int(cfg.DD3D.FCOS2D.INFERENCE.THRESH_WITH_CTR) and a.loc_offset_half == int(cfg.DD3D.FEATURE_LOCATIONS_OFFSET == "half")
    assert a.depth_is_distance == int(cfg.DD3D.FCOS3D.PREDICT_DISTANCE) and a.allocentric == int(cfg.DD3D.FCOS3D.PREDICT_ALLOCENTRIC_ROT)
    assert a.scale_depth_by_focal == int(cfg.DD3D.FCOS3D.SCALE_DEPTH_BY_FOCAL_LENGTHS) and a.class_agnostic_3d == int(C3 == 1)


@pytest.mark.parametrize("name", ["default"] + list(VARIANTS))
def test_predictor_folding_reproduces_the_reference_heads(hiplib, name):
    """The fused predictor launch on the CPU: packed filters (K order of the kernel), per-level Scale / Offset / bias folded into
    (scale, bias), ReLU as a per-channel lower clamp -- evaluated with plain torch on the oracle's tower outputs, every segment must
    give the reference head maps.  Pins the host-side folding for each head-construction variant without a GPU."""
    from dd3d_amd import META_ARCH_REGISTRY
    from dd3d_amd.engine import ConvOp, ForwardPlan
    from oracle import dd3d_oracle as O
    from tests.test_host_logic import _emulate_igemm
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", VARIANTS.get(name))
    model = META_ARCH_REGISTRY.get("DD3D")(cfg)
    model.load_state_dict(sd, strict=True)
    model.math = "bf16x3"  # the filter layout decoded below (three bf16 planes); the folding itself does not depend on the arithmetic
    plan = ForwardPlan(model, 1, 128, 256, device="cpu", dry_run=True)
    # (round 4: the groups of <= 32 channels -- cls, box2d + centerness -- run in their own launch on the 32-column tile, the wide box3d
    # group in another; their segments, in group order, are what round 3's single launch held)
    pops = [o for o in plan.ops if isinstance(o, ConvOp) and o.name in ("predictors.narrow", "predictors")]
    assert pops and pops[0].name == "predictors.narrow" and pops[0].info["tile"] == (128, 32) and len(pops) <= 2  # (class-agnostic box3d: 11 channels, all narrow)
    op = type("Segs", (), {"keep": [k for o in pops for k in o.keep]})()
    with torch.no_grad():
        _, st = O.dd3d_forward(sd, cfg, case_inputs(1, 128, 256, False, "kitti"), stop_after_heads=True)
        L = len(st["features"])
        towers = {
            0: [O._tower(sd, "fcos2d_head.cls_tower", f, l, cfg.DD3D.FCOS2D.NUM_CLS_CONVS) for l, f in enumerate(st["features"])],
            1: [O._tower(sd, "fcos2d_head.box2d_tower", f, l, cfg.DD3D.FCOS2D.NUM_BOX_CONVS) for l, f in enumerate(st["features"])],
            2: [O._tower(sd, "fcos3d_head.box3d_tower", f, l, cfg.DD3D.FCOS3D.NUM_CONVS) for l, f in enumerate(st["features"])],
        }
        want = {
            0: st["logits"], 1: [torch.cat([st["box2d_reg"][l], st["centerness"][l]], 1) for l in range(L)],
            2: [torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1) for l in range(L)],
        }
        meta = dict(KH=3, KW=3, Cin=256, Kpad=2304)
        assert len(op.keep) == 4 * 3 * L
        for grp in range(3):
            for l in range(L):
                w, scale, bias, lo = op.keep[4 * (grp * L + l):4 * (grp * L + l) + 4]
                n = want[grp][l].shape[1]
                if w.dim() == 4:  # split-operand layout Wp3[n][k-tile][hi | mid | lo][32] (bf16): the three planes sum to the f32 filter exactly
                    assert w.dtype == torch.int16 and w.shape[2:] == (3, 32)  # bf16 bit patterns
                    w = w.view(torch.bfloat16).float().sum(2).reshape(w.shape[0], -1)
                y = _emulate_igemm(towers[grp][l], w, dict(meta, N=n), 1, 1)
                y = y * scale[:n].view(1, -1, 1, 1) + bias[:n].view(1, -1, 1, 1)
                if lo is not None:
                    y = torch.maximum(y, lo[:n].view(1, -1, 1, 1))
                ref = want[grp][l]
                assert float((y - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), (grp, l)
