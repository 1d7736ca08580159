"""Edge case: images on which nothing passes PRE_NMS_THRESH.  The reference's own DD3D.forward returns empty `Instances` for them
(checked once with the reference class under the shims of tests/golden/ref_shims.py: len 0, pred_boxes (0, 4), int64 classes,
quat (0, 4), image_size kept); the oracle, the host-side collect and the HIP path must do the same.  (Named to run last.)"""
import types

import pytest
import torch

EMPTY = {"DD3D": {"FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.9999}}}}


def _host_plan(det_count, **fields):
    """What DD3D.collect reads of a plan, on the host: the detection buffer, K^-1 and the forward's read-back record
    (engine.PlanBase.readback: counts + status words in one copy)."""
    rb = types.SimpleNamespace(status=0, counts=det_count, amax=torch.zeros(0), flags=torch.zeros((0, 2), dtype=torch.int32))
    return types.SimpleNamespace(det_count=det_count, readback=lambda: rb, check_status=lambda rb=None: None, **fields)


def test_oracle_returns_empty_results():
    from dd3d_amd.synthetic import make_inputs
    from oracle import dd3d_oracle as O
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", EMPTY)
    with torch.no_grad():
        res, st = O.dd3d_forward(sd, cfg, make_inputs(2, 128, 256))
    assert all(len(info[i]["fg_inds"]) == 0 for info in st["level_info"] for i in range(2))
    for r in res:
        assert r["pred_boxes"].shape == (0, 4) and r["scores"].shape == (0, ) and r["pred_classes"].dtype == torch.int64
        assert r["pred_boxes3d"]["quat"].shape == (0, 4)


def test_collect_of_an_empty_detection_buffer(kitti_dla34):
    """Host side of the forward on a detection buffer with zero counts (no GPU needed: collect only reads plan tensors)."""
    from dd3d_amd import hip
    from dd3d_amd.synthetic import make_inputs
    _, model, _ = kitti_dla34
    B, cap = 2, 8
    plan = _host_plan(torch.zeros(B, dtype=torch.int32), det_cap=cap, det=torch.zeros(B, cap, hip.DET_FIELDS),
                      inv_K=torch.eye(3).reshape(1, 9).repeat(B, 1), has_global_boxes=False)
    inputs = make_inputs(B, 128, 256, out_hw=(99, 201))
    out = model.collect(plan, inputs, [(128, 256)] * B)
    for o in out:
        i = o["instances"]
        assert len(i) == 0 and tuple(i.image_size) == (99, 201)
        assert i.pred_boxes.tensor.shape == (0, 4) and i.scores.shape == (0, ) and i.pred_classes.dtype == torch.int64
        assert i.pred_boxes3d.quat.shape == (0, 4) and i.pred_boxes3d.vectorize().shape == (0, 10) and i.scores_3d.shape == (0, )


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("use_graph", [False, True])
def test_hip_forward_with_no_candidates(hiplib, use_graph):
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle, gpu_model
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", EMPTY)
    model = gpu_model(cfg, sd, use_graph=use_graph)
    inputs = make_inputs(2, 128, 256, out_hw=(99, 201))
    for _ in range(2):  # second call: replay
        out = model(inputs)
    plan = model.get_plan(2, 128, 256)
    assert plan.counts.cpu().abs().sum().item() == 0 and plan.det_count.cpu().abs().sum().item() == 0
    for o in out:
        i = o["instances"]
        assert len(i) == 0 and tuple(i.image_size) == (99, 201)
        assert i.pred_boxes.tensor.shape == (0, 4) and i.pred_classes.dtype == torch.int64 and i.pred_boxes3d.quat.shape == (0, 4)


def test_collect_does_not_alias_the_detection_buffer(kitti_dla34):
    """An image with exactly ONE detection: a (1, k) slice of the detection buffer counts as contiguous whatever its row stride, so
    `.contiguous()` alone would hand out views of the plan buffer, and the next forward on the same cached plan (TTA runs several
    batches through one plan before reading any output) would rewrite results already returned."""
    from dd3d_amd import hip
    from dd3d_amd.synthetic import make_inputs
    _, model, _ = kitti_dla34
    B, cap = 2, 8
    det = torch.arange(B * cap * hip.DET_FIELDS, dtype=torch.float32).reshape(B, cap, hip.DET_FIELDS)
    plan = _host_plan(torch.tensor([1, 3], dtype=torch.int32), det_cap=cap, det=det,
                      inv_K=torch.eye(3).reshape(1, 9).repeat(B, 1), has_global_boxes=False)
    inputs = make_inputs(B, 128, 256)
    out = model.collect(plan, inputs, [(128, 256)] * B)
    snap = [(o["instances"].pred_boxes.tensor.clone(), o["instances"].scores.clone(), o["instances"].scores_3d.clone(),
             o["instances"].locations.clone(), o["instances"].pred_boxes3d.quat.clone(), o["instances"].pred_boxes3d.size.clone()) for o in out]
    storage = det.untyped_storage().data_ptr()
    det.add_(1000.0)  # "the next forward" rewrites the plan buffer
    plan.inv_K.mul_(3.0)
    for o, s in zip(out, snap):
        i = o["instances"]
        for got, want in zip((i.pred_boxes.tensor, i.scores, i.scores_3d, i.locations, i.pred_boxes3d.quat, i.pred_boxes3d.size), s):
            assert got.untyped_storage().data_ptr() != storage
            assert torch.equal(got, want)
        assert torch.equal(i.pred_boxes3d.inv_intrinsics[0], torch.eye(3))
