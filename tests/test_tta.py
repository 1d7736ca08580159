"""TTA wrapper: oracle vs the golden produced by the reference's own DD3DWithTTA (CPU); HIP wrapper vs the same golden (GPU)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_tta_golden import TTA_OVERRIDES, tta_case

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tta_dla34.npz"))


def _bundle():
    from tests.util import bundle
    return bundle("dd3d_kitti_dla34", "dla34_kitti", TTA_OVERRIDES)


def _check(boxes, scores_3d, classes, vec, proj_ctr, tol):
    assert len(boxes) == len(G["boxes"]) == 192
    assert np.array_equal(np.asarray(classes), G["classes"])
    assert np.allclose(boxes, G["boxes"], rtol=tol, atol=tol * 300) and np.allclose(scores_3d, G["scores_3d"], rtol=tol, atol=1e-6)
    assert np.allclose(vec[:, 4:], G["vectorize"][:, 4:], rtol=tol, atol=tol * 80) and np.allclose(proj_ctr, G["proj_ctr"], rtol=tol, atol=tol * 300)
    q, gq = vec[:, :4], G["vectorize"][:, :4]
    assert float(np.minimum(np.abs(q - gq).max(1), np.abs(q + gq).max(1)).max()) < max(tol, 1e-5) * 10


def test_tta_oracle_matches_reference_golden():
    from dd3d_amd.structures import Pose
    from oracle import tta_oracle as T
    cfg, sd = _bundle()
    x = tta_case()
    x["extrinsics"] = Pose()
    with torch.no_grad():
        r = T.tta_forward(sd, cfg, x)
    assert r["n_union"] > len(G["boxes"])  # the merge NMS really removed something
    _check(r["pred_boxes"].numpy(), r["scores_3d"].numpy(), r["pred_classes"].numpy(), r["vec"].numpy(), r["proj_ctr"].numpy(), 1e-5)


def test_tta_transform_inverse_roundtrip():
    from dd3d_amd.tta import TTATransform
    t = TTATransform(None, (110, 260, 128, 303), 303)
    K = np.float32([[150.0, 0, 131.0], [0, 210.0, 55.0], [0, 0, 1]])
    assert np.allclose(t.inverse_intrinsics(t.apply_intrinsics(K)), K, rtol=1e-6)
    b = np.float32([[10, 5, 60, 40]])
    fwd = b.copy()
    fwd[:, [0, 2]] = fwd[:, [0, 2]] * (303 / 260)
    fwd[:, [1, 3]] = fwd[:, [1, 3]] * (128 / 110)
    fwd = np.float32([[303 - fwd[0, 2], fwd[0, 1], 303 - fwd[0, 0], fwd[0, 3]]])
    assert np.allclose(t.inverse_box(fwd), b, rtol=1e-5)
    v = np.float32([[0.9, 0.1, 0.3, 0.2, 1.5, 0.5, 12.0, 1.6, 3.9, 1.5]])
    assert np.allclose(t.inverse_box3d(t.inverse_box3d(v)), v)  # the mirror is an involution


@pytest.mark.gpu
def test_hip_tta_matches_reference_golden(hiplib):
    from dd3d_amd.structures import Pose
    from dd3d_amd.tta import DD3DWithTTA
    from tests.util import gpu_model
    cfg, sd = _bundle()
    model = gpu_model(cfg, sd, use_graph=False)
    tta = DD3DWithTTA(cfg, model)
    x = tta_case()
    x["extrinsics"] = Pose()
    inst = tta([x])[0]["instances"]
    assert tuple(inst.image_size) == (110, 260)
    _check(inst.pred_boxes.tensor.cpu().numpy(), inst.scores_3d.cpu().numpy(), inst.pred_classes.cpu().numpy(),
           inst.pred_boxes3d.vectorize().cpu().numpy(), inst.pred_boxes3d.proj_ctr.cpu().numpy(), 1e-3)
    with pytest.raises(AssertionError, match="postprocess_in_inference"):
        model.postprocess_in_inference = True
        DD3DWithTTA(cfg, model)


# ------------------------------------------------------------------------------------------------ NuscenesDD3DWithTTA
GN = np.load(os.path.join(os.path.dirname(__file__), "golden", "tta_nusc_dla34.npz"))


def _nusc_bundle():
    from tests.golden.make_tta_golden import NUSC_TTA_OVERRIDES
    from tests.util import bundle
    return bundle("dd3d_nusc_dla34", "dla34_nusc", NUSC_TTA_OVERRIDES)


def _check_nusc(i, boxes, scores_3d, classes, attrs, speeds, vec, glob, tol):
    n = int(GN[f"n{i}"])
    assert len(boxes) == n, (i, len(boxes), n)
    if n == 0:
        return
    assert np.array_equal(classes, GN[f"classes{i}"]) and np.array_equal(attrs, GN[f"attributes{i}"])
    assert np.allclose(boxes, GN[f"boxes{i}"], rtol=tol, atol=tol * 200) and np.allclose(scores_3d, GN[f"scores_3d{i}"], rtol=tol, atol=1e-6)
    assert np.allclose(speeds, GN[f"speeds{i}"], rtol=tol, atol=1e-5) and np.allclose(vec[:, 4:], GN[f"vectorize{i}"][:, 4:], rtol=tol, atol=tol * 80)
    g, gg = glob, GN[f"global{i}"]
    assert np.allclose(g[:, 4:], gg[:, 4:], rtol=tol, atol=max(tol, 2e-7) * 1500)  # world-frame translations ~1e3 m
    assert float(np.minimum(np.abs(g[:, :4] - gg[:, :4]).max(1), np.abs(g[:, :4] + gg[:, :4]).max(1)).max()) < max(tol, 1e-5) * 10


def test_nuscenes_tta_oracle_matches_reference_golden():
    from oracle import dd3d_oracle as O
    from oracle import tta_oracle as T
    from tests.golden.make_tta_golden import nusc_tta_case
    cfg, sd = _nusc_bundle()
    with torch.no_grad():
        out, merged = T.nuscenes_tta_forward(sd, cfg, nusc_tta_case())
    assert sum(len(m["scores"]) for m in merged) > sum(int(GN[f"n{i}"]) for i in range(6)) == 120  # aggregation + the 120 cap act
    for i, r in enumerate(out):
        _check_nusc(i, r["pred_boxes"].numpy(), r["scores_3d"].numpy(), r["pred_classes"].numpy(), r["pred_attributes"].numpy(), r["pred_speeds"].numpy(),
                    O.boxes3d_vectorize(r["pred_boxes3d"]).numpy(), r["pred_boxes3d_global"].numpy(), 1e-5)


@pytest.mark.gpu
def test_hip_nuscenes_tta_matches_reference_golden(hiplib):
    from dd3d_amd.tta import NuscenesDD3DWithTTA
    from tests.golden.make_tta_golden import nusc_tta_case
    from tests.util import gpu_model
    cfg, sd = _nusc_bundle()
    model = gpu_model(cfg, sd, use_graph=False)
    res = NuscenesDD3DWithTTA(cfg, model)(nusc_tta_case())
    assert len(res) == 6
    for i, r in enumerate(res):
        o = r["instances"]
        assert tuple(o.image_size) == (100, 178)
        _check_nusc(i, o.pred_boxes.tensor.cpu().numpy(), o.scores_3d.cpu().numpy(), o.pred_classes.cpu().numpy(), o.pred_attributes.cpu().numpy(),
                    o.pred_speeds.cpu().numpy(), o.pred_boxes3d.vectorize().cpu().numpy(), o.pred_boxes3d_global.vectorize().cpu().numpy(), 1e-3)


# ------------------------------------------------------------------------------------------------ the experiment's own TTA, at its scales
GF = np.load(os.path.join(os.path.dirname(__file__), "golden", "tta_dla34_kitti_scales.npz"))


def _full_bundle():
    from tests.golden.make_tta_golden import FULL_TTA_OVERRIDES
    from tests.util import bundle
    return bundle("dd3d_kitti_dla34", "dla34_kitti", FULL_TTA_OVERRIDES)


def _match_full(boxes, scores_3d, classes, vec, tol, max_unmatched=0):
    """Detections matched to the golden's by class and 2D box (the merged list is ranked by scores_3d: two near-equal scores may swap
    places without any field being wrong); every matched detection within `tol`; at most `max_unmatched` on either side unmatched."""
    gb, gs, gc, gv = GF["boxes"], GF["scores_3d"], GF["classes"], GF["vectorize"]
    used, pairs = set(), []
    for i in range(len(boxes)):
        cand = [j for j in np.nonzero(gc == classes[i])[0] if j not in used and np.abs(gb[j] - boxes[i]).max() <= max(tol * 2000, 0.05)]
        if cand:
            j = min(cand, key=lambda j: np.abs(gb[j] - boxes[i]).max())
            used.add(j)
            pairs.append((i, j))
    unmatched = (len(boxes) - len(pairs)) + (len(gb) - len(pairs))
    assert unmatched <= 2 * max_unmatched, (len(boxes), len(gb), len(pairs))
    ih, ig = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    assert np.allclose(scores_3d[ih], gs[ig], rtol=tol, atol=1e-6)
    assert np.allclose(vec[ih, 4:], gv[ig, 4:], rtol=tol, atol=tol * 80)
    q, gq = vec[ih, :4], gv[ig, :4]
    assert float(np.minimum(np.abs(q - gq).max(1), np.abs(q + gq).max(1)).max()) < max(tol, 1e-5) * 10
    return len(pairs), unmatched


def test_tta_oracle_matches_reference_golden_at_the_experiments_scales():
    """configs/experiments/dd3d_kitti_dla34.yaml:44-53 -- MIN_SIZES [320, 384, 448, 512, 576] x flip on one 370 x 1224 frame, the ten
    copies forwarded as ONE batch of IMS_PER_BATCH (80) on a 640 x 1920 canvas (test_time_augmentation.py:59-66,118-133), merged by the
    class-aware NMS (:163-181): the oracle against the reference's own DD3DWithTTA (tests/golden/make_tta_golden.py full)."""
    from dd3d_amd.structures import Pose
    from oracle import tta_oracle as T
    from tests.golden.make_tta_golden import full_tta_case
    cfg, sd = _full_bundle()
    assert list(cfg.TEST.AUG.MIN_SIZES) == [320, 384, 448, 512, 576] and cfg.TEST.AUG.FLIP and cfg.TEST.IMS_PER_BATCH == 80
    assert [tuple(s) for s in GF["copy_shapes"]] == [(320, 1059)] * 2 + [(384, 1270)] * 2 + [(448, 1482)] * 2 + [(512, 1694)] * 2 + [(576, 1905)] * 2
    x = full_tta_case()
    x["extrinsics"] = Pose()
    with torch.no_grad():
        r = T.tta_forward(sd, cfg, x)
    assert r["n_union"] == int(GF["per_copy"].sum()) and r["n_union"] > len(GF["boxes"]) == 390
    assert np.array_equal(r["pred_classes"].numpy(), GF["classes"])
    n, un = _match_full(r["pred_boxes"].numpy(), r["scores_3d"].numpy(), r["pred_classes"].numpy(), r["vec"].numpy(), 1e-5)
    assert (n, un) == (390, 0)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_hip_tta_matches_reference_golden_at_the_experiments_scales(hiplib):
    """The HIP wrapper on the same frame: five device-side Pillow-exact resizes, ten copies in ONE launch plan (B = 10, 640 x 1920), inverse
    maps, merge NMS by dd3d_nms_finalize -- against the reference's own merged detections.  A candidate that sits ON a selection cut of one
    of the ten forwards may flip (bounded here, not waved through: <= 2 of 390)."""
    from dd3d_amd.structures import Pose
    from dd3d_amd.tta import DD3DWithTTA
    from tests.golden.make_tta_golden import full_tta_case
    from tests.util import gpu_model
    cfg, sd = _full_bundle()
    model = gpu_model(cfg, sd, use_graph=True)
    tta = DD3DWithTTA(cfg, model)
    assert tta.batch_size == 80
    x = full_tta_case()
    x["extrinsics"] = Pose()
    inst = tta([x])[0]["instances"]
    assert tuple(inst.image_size) == (370, 1224)
    plan = next(iter(model._plans.values()))
    assert (plan.B, plan.Hp, plan.Wp) == (10, 640, 1920)  # the reference's ImageList of the ten copies (size_divisibility 128)
    n, un = _match_full(inst.pred_boxes.tensor.cpu().numpy(), inst.scores_3d.cpu().numpy(), inst.pred_classes.cpu().numpy(),
                        inst.pred_boxes3d.vectorize().cpu().numpy(), 1e-3, max_unmatched=2)
    print(f"[tta full] {n} of 390 merged detections matched, {un} unmatched")
    assert n >= 388


# ------------------------------------------------------------------------------------------------ the nuScenes experiment's own TTA, at its scales
GNF = np.load(os.path.join(os.path.dirname(__file__), "golden", "tta_nusc_dla34_scales.npz"))


def _nusc_full_bundle():
    from tests.golden.make_tta_golden import NUSC_FULL_TTA_OVERRIDES
    from tests.util import bundle
    return bundle("dd3d_nusc_dla34", "dla34_nusc", NUSC_FULL_TTA_OVERRIDES)


def _match_nusc_full(i, boxes, scores_3d, classes, attrs, speeds, vec, glob, tol, max_unmatched=0):
    """Camera i's detections matched to the golden's by class and 2D box (the per-camera lists are ranked by score: near-equal scores may
    swap places without any field being wrong); every matched detection within `tol`, attributes exact; at most `max_unmatched` on either
    side unmatched.  Returns (matched, unmatched)."""
    n = int(GNF[f"n{i}"])
    gb, gs, gc = GNF[f"boxes{i}"], GNF[f"scores_3d{i}"], GNF[f"classes{i}"]
    used, pairs = set(), []
    for a in range(len(boxes)):
        cand = [j for j in np.nonzero(gc == classes[a])[0] if j not in used and np.abs(gb[j] - boxes[a]).max() <= max(tol * 2000, 0.05)]
        if cand:
            j = min(cand, key=lambda j: np.abs(gb[j] - boxes[a]).max())
            used.add(j)
            pairs.append((a, j))
    unmatched = (len(boxes) - len(pairs)) + (n - len(pairs))
    assert unmatched <= 2 * max_unmatched, (i, len(boxes), n, len(pairs))
    ih, ig = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    assert np.array_equal(attrs[ih], GNF[f"attributes{i}"][ig])
    assert np.allclose(scores_3d[ih], gs[ig], rtol=tol, atol=1e-6) and np.allclose(speeds[ih], GNF[f"speeds{i}"][ig], rtol=tol, atol=1e-5)
    gv, gg = GNF[f"vectorize{i}"], GNF[f"global{i}"]
    assert np.allclose(vec[ih, 4:], gv[ig, 4:], rtol=tol, atol=tol * 80)
    assert np.allclose(glob[ih, 4:], gg[ig, 4:], rtol=tol, atol=max(tol, 2e-7) * 1500)  # world-frame translations ~1e3 m
    for q, gq in ((vec[ih, :4], gv[ig, :4]), (glob[ih, :4], gg[ig, :4])):
        assert float(np.minimum(np.abs(q - gq).max(1), np.abs(q + gq).max(1)).max()) < max(tol, 1e-5) * 10
    return len(pairs), unmatched


def test_nuscenes_tta_golden_is_the_experiments_configuration():
    """What tests/golden/make_tta_golden.py `nusc full` recorded of the reference's own NuscenesDD3DWithTTA (nuscenes_dd3d_tta.py:21-178) under
    configs/experiments/dd3d_nusc_dla34.yaml:55-62: MIN_SIZES [640 ... 1152] x flip on 900 x 1600 frames, ten copies per camera in one batch
    (IMS_PER_BATCH 96), the per-sample cap of 500 reached by the aggregation."""
    cfg, _ = _nusc_full_bundle()
    assert list(cfg.TEST.AUG.MIN_SIZES) == [640, 768, 896, 1024, 1152] and cfg.TEST.AUG.FLIP and cfg.TEST.IMS_PER_BATCH == 96
    assert int(GNF["batch_size"]) == 96
    assert [tuple(s) for s in GNF["copy_shapes"]] == [(640, 1138)] * 2 + [(768, 1365)] * 2 + [(896, 1593)] * 2 + [(1024, 1820)] * 2 + [(1152, 2048)] * 2
    n = [int(GNF[f"n{i}"]) for i in range(6)]
    assert sum(n) == 500 == cfg.DD3D.NUSC.INFERENCE.MAX_NUM_DETS_PER_SAMPLE and int(GNF["merged_per_image"].sum()) > 500  # the cap acts


@pytest.mark.skipif(os.environ.get("DD3D_SLOW_TESTS", "0") != "1", reason="~6 min and ~30 GB of CPU: the oracle's sixty 1152 x 2048-canvas forwards (DD3D_SLOW_TESTS=1)")
def test_nuscenes_tta_oracle_matches_reference_golden_at_the_experiments_scales():
    from oracle import dd3d_oracle as O
    from oracle import tta_oracle as T
    from tests.golden.make_tta_golden import nusc_full_tta_case
    cfg, sd = _nusc_full_bundle()
    with torch.no_grad():
        out, merged = T.nuscenes_tta_forward(sd, cfg, nusc_full_tta_case())
    assert [len(m["scores"]) for m in merged] == GNF["merged_per_image"].tolist()
    for i, r in enumerate(out):
        n, un = _match_nusc_full(i, r["pred_boxes"].numpy(), r["scores_3d"].numpy(), r["pred_classes"].numpy(), r["pred_attributes"].numpy(),
                                 r["pred_speeds"].numpy(), O.boxes3d_vectorize(r["pred_boxes3d"]).numpy(), r["pred_boxes3d_global"].numpy(), 1e-5)
        assert (n, un) == (int(GNF[f"n{i}"]), 0)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_hip_nuscenes_tta_matches_reference_golden_at_the_experiments_scales(hiplib):
    """The HIP wrapper on the same sample (round-5 verdict: the nuScenes wrapper was only checked at 96-128 px): per camera five device-side
    Pillow-exact resizes of the 900 x 1600 frame, the ten copies in ONE launch plan (B = 10 on the 1152 x 2048 canvas, as the reference's
    ImageList pads them), inverse maps, the per-camera merge NMS, then the sample-level BEV aggregation with the 500-box cap -- against the
    reference's own 500 detections.  A candidate ON a selection cut of one of the sixty forwards may flip and displace a neighbour of the
    capped, score-ordered sample list: bounded per camera (<= 3 of ~83), not waved through."""
    from dd3d_amd.tta import NuscenesDD3DWithTTA
    from tests.golden.make_tta_golden import nusc_full_tta_case
    from tests.util import gpu_model
    cfg, sd = _nusc_full_bundle()
    model = gpu_model(cfg, sd, use_graph=True)
    tta = NuscenesDD3DWithTTA(cfg, model)
    assert tta.batch_size == 96
    res = tta(nusc_full_tta_case())
    assert len(res) == 6
    plan = next(iter(model._plans.values()))
    assert (plan.B, plan.Hp, plan.Wp) == (10, 1152, 2048)
    matched = unmatched = 0
    for i, r in enumerate(res):
        o = r["instances"]
        assert tuple(o.image_size) == (900, 1600)
        n, un = _match_nusc_full(i, o.pred_boxes.tensor.cpu().numpy(), o.scores_3d.cpu().numpy(), o.pred_classes.cpu().numpy(),
                                 o.pred_attributes.cpu().numpy(), o.pred_speeds.cpu().numpy(), o.pred_boxes3d.vectorize().cpu().numpy(),
                                 o.pred_boxes3d_global.vectorize().cpu().numpy(), 1e-3, max_unmatched=3)
        matched, unmatched = matched + n, unmatched + un
    print(f"[nusc tta full] {matched} of 500 detections matched, {unmatched} unmatched")
    assert matched >= 494
