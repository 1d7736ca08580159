"""Backbone / pyramid variants outside the reference's experiment configs, on the GPU: forward vs the oracle (which is pinned against the
reference for each of them, tests/test_plan_emulation.py).  Their host side already equals the oracle in the CPU plan emulation; these run
the same kernels on the variants' shapes."""
import pytest
import torch

CASES = {
    "dla60": ("dd3d_kitti_dla34", "dla60_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-60"}}}, 128, 256),
    "dla102": ("dd3d_kitti_dla34", "dla102_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-102"}}}, 128, 256),
    "dlax46c": ("dd3d_kitti_dla34", "dlax46c_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-46-C"}}}, 128, 256),
    "v19": ("dd3d_kitti_v99", "v99_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-eSE"}}}, 64, 128),
    "v19slim": ("dd3d_kitti_v99", "v19slim_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-slim-eSE"}}}, 64, 128),  # padded 80- / 112-channel slices
    "v19slimdw": ("dd3d_kitti_v99", "v19slimdw_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-slim-dw-eSE"}}}, 64, 128),  # + depthwise layers
    "three_levels": ("dd3d_kitti_dla34", "dla34_kitti", {"DD3D": {"IN_FEATURES": ["p3", "p4", "p5"]}}, 128, 256),
}


@pytest.mark.gpu
@pytest.mark.timeout(240)
@pytest.mark.parametrize("name", list(CASES))
def test_hip_variant_forward_matches_oracle(hiplib, name):
    from dd3d_amd.synthetic import make_inputs
    from tests.test_forward_gpu import _check_final, _check_head_maps, _oracle
    from tests.util import bundle, gpu_model, oracle_heads_to_plan
    exp, tag, over, H, W = CASES[name]
    cfg, sd = bundle(exp, tag, over)
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(2, H, W)
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    _check_head_maps(plan, st, C)
    oracle_heads_to_plan(plan, st, C)  # integer parity on identical head maps
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    for i in range(2):
        _check_final(out[i], ref[i])
