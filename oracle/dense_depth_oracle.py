"""CPU ORACLE for DD3DDenseDepth (test infrastructure only; rules in dd3d_oracle.py).

Restates tridet/modeling/dd3d/dense_depth.py:89-100 (head: box3d tower -> per-level 1-channel predictor -> Scale -> Offset) and
:121-151 (forward up to the losses: aligned_bilinear of every level by its stride, tridet/utils/tensor2d.py:28-47, then division
by the focal-length pixel size).  Pinned by tests/golden/dla34_densedepth_*.npz, recorded from the reference class itself.
"""
import torch
import torch.nn.functional as F

from oracle import dd3d_oracle as O


def aligned_bilinear(t, factor, offset="none"):
    """tensor2d.py:28-47."""
    if factor == 1:
        return t
    h, w = t.shape[2:]
    t = F.pad(t, pad=(0, 1, 0, 1), mode="replicate")
    oh, ow = factor * h + 1, factor * w + 1
    t = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=True)
    if offset == "half":
        t = F.pad(t, pad=(factor // 2, 0, factor // 2, 0), mode="replicate")
    return t[:, :, :oh - 1, :ow - 1]


def dense_depth_forward(sd, cfg, batched_inputs):
    """Returns (list over levels of (B, Hp, Wp) depth maps, stages)."""
    c3 = cfg["DD3D"]["FCOS3D"]
    x, image_sizes, intrinsics = O.preprocess(sd, batched_inputs, O.size_divisibility(cfg))
    features, strides, _ = O.dd3d_backbone(sd, cfg, x)
    raw = []
    for l, f in enumerate(features):
        t = O._tower(sd, "fcos3d_head.box3d_tower", f, l, c3["NUM_CONVS"])
        d = O.conv2d(sd, f"fcos3d_head.dense_depth.{l}", t, padding=1)
        if c3["USE_SCALE"]:
            d = d * sd[f"fcos3d_head.scales_depth.{l}.scale"] + sd[f"fcos3d_head.offsets_depth.{l}.bias"]
        raw.append(d)
    maps = [aligned_bilinear(d, s, cfg["DD3D"]["FEATURE_LOCATIONS_OFFSET"]).squeeze(1) for d, s in zip(raw, strides)]
    if c3["SCALE_DEPTH_BY_FOCAL_LENGTHS"]:
        inv_K = intrinsics.inverse()
        pixel_size = torch.norm(torch.stack([inv_K[:, 0, 0], inv_K[:, 1, 1]], dim=-1), dim=-1)
        scaled = (pixel_size * c3["SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR"]).reshape(-1, 1, 1)
        maps = [m / scaled for m in maps]
    return maps, dict(images=x, features=features, raw=raw)
