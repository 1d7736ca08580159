"""DD3DDenseDepth: oracle vs the golden recorded from the reference class (CPU), HIP path vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dla34_densedepth_128x256_b2.npz")
OVER = {"MODEL": {"META_ARCHITECTURE": "DD3DDenseDepth"}, "DD3D": {"IN_FEATURES": ["p3", "p4", "p5", "p6", "p7"]}}


def _case():
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", OVER)
    inputs = make_inputs(2, 128, 256)
    inputs[1]["intrinsics"] = inputs[1]["intrinsics"] * torch.tensor([[1.25], [1.25], [1.0]])
    return cfg, sd, inputs


def test_dense_depth_oracle_matches_reference_golden():
    from oracle import dense_depth_oracle as D
    cfg, sd, inputs = _case()
    assert cfg.MODEL.META_ARCHITECTURE == "DD3DDenseDepth" and any(k.startswith("fcos3d_head.dense_depth.4.") for k in sd)
    g = np.load(GOLD)
    with torch.no_grad():
        maps, _ = D.dense_depth_forward(sd, cfg, inputs)
    assert len(maps) == 5
    for l, m in enumerate(maps):
        ref = torch.from_numpy(g[f"depth{l}"])
        assert m.shape == ref.shape == (2, 128, 256)
        assert torch.allclose(m, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max())), l


def test_aligned_bilinear_properties():
    """factor 1 is the identity; sample points on the coarse grid reproduce the input; offset 'half' shifts by factor // 2."""
    from oracle.dense_depth_oracle import aligned_bilinear
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 1, 5, 7, generator=g)
    assert torch.equal(aligned_bilinear(x, 1), x)
    y = aligned_bilinear(x, 4)
    assert y.shape == (1, 1, 20, 28) and torch.allclose(y[:, :, ::4, ::4], x, atol=1e-6)
    z = aligned_bilinear(x, 4, "half")
    assert z.shape == y.shape and torch.allclose(z[:, :, 2:, 2:], y[:, :, :-2, :-2], atol=1e-6)


def test_eval_forward_raises_like_the_reference():
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    model = META_ARCH_REGISTRY.get("DD3DDenseDepth")(get_cfg("dd3d_kitti_dla34", OVER))
    with pytest.raises(NotImplementedError):
        model([])  # dense_depth.py:162-163


@pytest.mark.gpu
@pytest.mark.parametrize("math", [None, "bf16x3", "f32"], ids=["f16x2", "bf16x3", "f32mfma"])
def test_hip_dense_depth_matches_oracle_and_golden(hiplib, math):
    from oracle import dense_depth_oracle as D
    from tests.util import gpu_model, max_abs
    cfg, sd, inputs = _case()
    model = gpu_model(cfg, sd, use_graph=True, math=math)
    maps = model.predict_dense_depth(inputs)
    maps2 = model.predict_dense_depth(inputs)  # graph replay
    with torch.no_grad():
        ref, st = D.dense_depth_forward(sd, cfg, inputs)
    g = np.load(GOLD)
    for l in range(5):
        tol = 1e-3 * float(ref[l].abs().max())  # north-star float bar; measured ~1e-5
        assert max_abs(maps[l], ref[l]) < tol and max_abs(maps[l], torch.from_numpy(g[f"depth{l}"])) < tol
        assert torch.equal(maps[l], maps2[l])
    # offset "half" variant of the upsampling, same weights
    from dd3d_amd import get_cfg
    cfg_h = get_cfg("dd3d_kitti_dla34", dict(OVER, DD3D=dict(OVER["DD3D"], FEATURE_LOCATIONS_OFFSET="half")))
    mh = gpu_model(cfg_h, sd, use_graph=False, math=math).predict_dense_depth(inputs)
    with torch.no_grad():
        rh, _ = D.dense_depth_forward(sd, cfg_h, inputs)
    for l in range(5):
        assert max_abs(mh[l], rh[l]) < 1e-3 * float(rh[l].abs().max())
