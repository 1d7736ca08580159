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
    assert len(boxes) == len(G["boxes"]) == 186
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
