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
@pytest.mark.xfail(strict=False, reason="written after the round's GPU minutes were spent: these switches have not run on hardware yet")
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
