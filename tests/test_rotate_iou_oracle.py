"""Evaluator-side overlap kernels: the oracle vs the golden vectors produced by the reference's own functions."""
import os

import numpy as np
import pytest

from oracle import rotate_iou_oracle as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotate_iou.npz"))


@pytest.mark.parametrize("crit", [-1, 0, 1, 2])
def test_rotate_iou_oracle_matches_reference(crit):
    got = R.rotate_iou_eval(G["boxes"], G["qboxes"], crit)
    ref = G[f"riou_{crit}"]
    assert (got > 0).sum() == (ref > 0).sum() > 150
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-6)
    if crit == -1:  # known answers: identical boxes, contained box, edge-touching boxes, far apart
        assert abs(ref[0, 0] - 1.0) < 1e-5 and abs(ref[1, 2] - 1.0 / 16.0) < 1e-5 and ref[2, 3] < 1e-4 and ref[0, 4] == 0.0


@pytest.mark.parametrize("crit,cam", [(-1, True), (-1, False), (0, True), (1, False)])
def test_d3_overlap_oracle_matches_reference(crit, cam):
    got = R.d3_box_overlap(G["boxes3d"], G["qboxes3d"], G["riou_2"], crit, cam)
    assert np.allclose(got, G[f"d3_{crit}_{int(cam)}"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("crit", [-1, 0, 1])
def test_image_overlap_oracle_matches_reference(crit):
    assert np.allclose(R.image_box_overlap(G["iboxes"], G["iqboxes"], crit), G[f"image_{crit}"], rtol=1e-6, atol=1e-7)
