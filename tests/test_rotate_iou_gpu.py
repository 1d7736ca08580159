"""HIP evaluator-side overlap kernels (through the C ABI) vs the golden vectors of the reference's own functions and vs the oracle
on a larger random problem."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotate_iou.npz"))


@pytest.mark.parametrize("crit", [-1, 0, 1, 2])
def test_rotate_iou_matches_reference_golden(hiplib, crit):
    from dd3d_amd.evaluators import rotate_iou_gpu_eval
    got = rotate_iou_gpu_eval(G["boxes"], G["qboxes"], crit)
    ref = G[f"riou_{crit}"]
    assert got.shape == ref.shape and (got > 0).sum() == (ref > 0).sum()
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_d3_and_image_overlap_match_reference_golden(hiplib):
    from dd3d_amd.evaluators import d3_box_overlap, d3_box_overlap_kernel, image_box_overlap
    for crit in (-1, 0, 1):
        for cam in (True, False):
            rinc = G["riou_2"].copy()
            d3_box_overlap_kernel(G["boxes3d"], G["qboxes3d"], rinc, crit, cam)
            assert np.allclose(rinc, G[f"d3_{crit}_{int(cam)}"], rtol=1e-4, atol=1e-5)
        assert np.allclose(image_box_overlap(G["iboxes"], G["iqboxes"], crit), G[f"image_{crit}"], rtol=1e-5, atol=1e-6)
    assert np.allclose(d3_box_overlap(G["boxes3d"], G["qboxes3d"], -1, True), G["d3_-1_1"], rtol=1e-4, atol=1e-5)


def test_rotate_iou_large_random_vs_oracle_and_properties(hiplib):
    from dd3d_amd.evaluators import rotate_iou_gpu_eval
    from oracle import rotate_iou_oracle as R
    from tests.golden.make_rotate_iou_golden import make_boxes
    rng = np.random.default_rng(11)
    boxes, q = make_boxes(rng, 300, spread=10.0), make_boxes(rng, 257, spread=10.0)
    got = rotate_iou_gpu_eval(boxes, q, -1)
    assert got.shape == (300, 257) and float(got.min()) >= 0.0 and float(got.max()) <= 1.0 + 1e-5
    sub = R.rotate_iou_eval(boxes[:40], q[:50], -1)
    assert np.allclose(got[:40, :50], sub, rtol=1e-4, atol=1e-5)
    # symmetry of the IoU, and IoU(b, b) == 1
    assert np.allclose(rotate_iou_gpu_eval(q, boxes, -1), got.T, rtol=1e-4, atol=1e-5)
    assert np.allclose(np.diag(rotate_iou_gpu_eval(boxes, boxes, -1)), 1.0, atol=1e-4)
    assert rotate_iou_gpu_eval(boxes[:0], q, -1).shape == (0, 257)
