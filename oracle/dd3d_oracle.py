"""CPU ORACLE for the DD3D inference forward path.  TEST INFRASTRUCTURE ONLY.

This file is a dependency-free (torch CPU fp32 only) restatement of the reference's
inference forward, written as plain functions over a flat ``state_dict``.  It is the
checker the HIP path is compared against; it is never imported by the product package
``dd3d_amd`` (only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg).

Every function cites the reference file:line (under /root/reference) it follows.

PARITY STATUS
-------------
* The in-repo part of the path (tridet/modeling/dd3d/*, feature_extractor/dla.py,
  structures/*, utils/geometry.py, utils/tensor2d.py, layers/normalization.py) is pinned
  by ``tests/golden/*.npz``: those vectors were produced by importing the *real* reference
  modules from /root/reference (see ``tests/golden/make_golden.py``) on top of small shims
  for the third-party packages that are not installed in this image.
* The third-party arithmetic the reference calls into -- detectron2 (Conv2d/FrozenBN/FPN/
  LastLevelP6P7/batched_nms/detector_postprocess, unpinned wheel for torch1.9 => v0.5/0.6),
  torchvision 0.10.0 (nms / batched_nms), pytorch3d (0.5.0-0.6.x, quaternion_to_matrix /
  matrix_to_quaternion) -- is NOT under /root/reference and cannot be imported here.  Its
  published algorithms are restated below (functions marked [ext]) => for those pieces
  **parity is unpinned**.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1e-7  # tridet/modeling/dd3d/fcos3d.py:13, tridet/utils/geometry.py:12


# ----------------------------------------------------------------------------------------
# [ext] detectron2 layers
# ----------------------------------------------------------------------------------------
def batch_norm_eval(sd, prefix, x, hook=None):
    """[ext] detectron2 ``get_norm("FrozenBN")`` = FrozenBatchNorm2d(eps=1e-5) and
    ``get_norm("BN")`` = nn.BatchNorm2d in eval mode.  Both evaluate
    ``F.batch_norm(x, running_mean, running_var, weight, bias, training=False, eps=1e-5)``
    under no_grad (FrozenBatchNorm2d.forward, non-grad branch).  ``hook`` lets the synthetic
    weight calibration (tests/golden/calibrate_synthetic.py) rewrite the statistics in
    ``sd`` from the pre-norm activation before they are applied."""
    if hook is not None:
        hook(prefix, x, sd)
    return F.batch_norm(
        x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
        training=False, eps=1e-5
    )


def conv2d(sd, name, x, stride=1, padding=0, norm=None, relu=False, hook=None):
    """[ext] detectron2 ``layers.Conv2d``: y = conv(x); y = norm(y); y = activation(y).
    ``norm`` is the state-dict prefix of the norm to apply (``<name>.norm`` or
    ``<name>.norm.<level>`` for a ModuleListDial, tridet/layers/normalization.py:30-40)."""
    w = sd[name + ".weight"]
    y = F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=padding, groups=x.shape[1] // w.shape[1])  # grouped: BottleneckX
    if norm is not None:
        y = batch_norm_eval(sd, norm, y, hook)
    if relu:
        y = F.relu(y)
    return y


# ----------------------------------------------------------------------------------------
# DLA-34  (tridet/modeling/feature_extractor/dla.py)
# ----------------------------------------------------------------------------------------
def _basic_block(sd, p, x, stride, residual=None, hook=None):
    """dla.py:24-62 BasicBlock: conv1(3x3,stride)+norm, relu, conv2(3x3)+norm, += residual, relu."""
    if residual is None:
        residual = x
    out = conv2d(sd, p + ".conv1", x, stride=stride, padding=1, norm=p + ".conv1.norm", relu=True, hook=hook)
    out = conv2d(sd, p + ".conv2", out, stride=1, padding=1, norm=p + ".conv2.norm", hook=hook)
    return F.relu(out + residual)


def _bottleneck(sd, p, x, stride, residual=None, hook=None):
    """dla.py:65-100 Bottleneck: conv1(1x1)+norm, relu, conv2(3x3,stride)+norm, relu, conv3(1x1)+norm, += residual, relu."""
    if residual is None:
        residual = x
    out = conv2d(sd, p + ".conv1", x, norm=p + ".conv1.norm", relu=True, hook=hook)
    out = conv2d(sd, p + ".conv2", out, stride=stride, padding=1, norm=p + ".conv2.norm", relu=True, hook=hook)
    out = conv2d(sd, p + ".conv3", out, norm=p + ".conv3.norm", hook=hook)
    return F.relu(out + residual)


def _root(sd, p, xs, hook=None, residual=False):
    """dla.py:146-167 Root: 1x1 conv over torch.cat(children, 1) + norm (+ children[0] iff residual) + relu."""
    y = conv2d(sd, p + ".conv", torch.cat(xs, 1), norm=p + ".conv.norm", hook=hook)
    return F.relu(y + xs[0] if residual else y)


def _tree(sd, p, x, levels, in_ch, out_ch, stride, level_root, children=None, hook=None, block=_basic_block, root_residual=False):
    """dla.py:170-247 Tree.forward.  ``project`` exists iff in!=out and tree1 is a block, not a Tree
    (dla.py:227-231); ``downsample`` = MaxPool2d(stride) iff stride>1 (dla.py:224-225)."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride=stride) if stride > 1 else x
    if in_ch != out_ch and levels == 1:
        residual = conv2d(sd, p + ".project", bottom, norm=p + ".project.norm", hook=hook)
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = block(sd, p + ".tree1", x, stride, residual, hook=hook)
        x2 = block(sd, p + ".tree2", x1, 1, None, hook=hook)
        return _root(sd, p + ".root", [x2, x1] + children, hook=hook, residual=root_residual)
    # levels > 1: tree1 is itself a Tree and ignores `residual` (dla.py:236-238 comment).
    x1 = _tree(sd, p + ".tree1", x, levels - 1, in_ch, out_ch, stride, False, None, hook=hook, block=block, root_residual=root_residual)
    children.append(x1)
    return _tree(sd, p + ".tree2", x1, levels - 1, out_ch, out_ch, 1, False, children, hook=hook, block=block, root_residual=root_residual)


DLA34_LEVELS = [1, 1, 1, 2, 2, 1]  # dla.py:359-361
DLA34_CHANNELS = [16, 32, 64, 128, 256, 512]
# name -> (levels, channels, block, residual_root): dla.py:359-427
DLA_SPECS = {
    "DLA-34": (DLA34_LEVELS, DLA34_CHANNELS, _basic_block, False),
    "DLA-46-C": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], _bottleneck, False),
    "DLA-60": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, False),
    "DLA-102": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, True),
    "DLA-169": ([1, 1, 2, 3, 5, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, True),
    # BottleneckX (dla.py:103-143) = Bottleneck with `planes * cardinality / 32` inner channels and a grouped 3x3; conv2d reads the group
    # count off the filter shape
    "DLA-X-46-C": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], _bottleneck, False),
    "DLA-X-60-C": ([1, 1, 1, 2, 3, 1], [16, 32, 64, 64, 128, 256], _bottleneck, False),
    "DLA-X-60": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, False),
    "DLA-X-102": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, True),
    "DLA-X-102-64": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], _bottleneck, True),
}


def dla_forward(sd, x, name="DLA-34", out_features=("level3", "level4", "level5"), prefix="backbone.bottom_up", hook=None):
    """dla.py:346-355 DLA.forward.  Layer construction: base_layer 7x7 s1 p3 (dla.py:271-280), level0/level1 = _make_conv_level
    (dla.py:327-344: levels[i] convs, the first one strided), level2..5 = Tree (dla.py:283-294)."""
    p = prefix
    levels, ch, block, residual_root = DLA_SPECS[name]
    outs = OrderedDict()
    x = conv2d(sd, p + ".base_layer", x, padding=3, norm=p + ".base_layer.norm", relu=True, hook=hook)
    for i in range(levels[0]):
        x = conv2d(sd, f"{p}.level0.{i}", x, padding=1, norm=f"{p}.level0.{i}.norm", relu=True, hook=hook)
    outs["level0"] = x
    for i in range(levels[1]):
        x = conv2d(sd, f"{p}.level1.{i}", x, stride=2 if i == 0 else 1, padding=1, norm=f"{p}.level1.{i}.norm", relu=True, hook=hook)
    outs["level1"] = x
    for lvl in range(2, 6):
        x = _tree(sd, f"{p}.level{lvl}", x, levels[lvl], ch[lvl - 1], ch[lvl], 2, level_root=(lvl >= 3), hook=hook, block=block,
                  root_residual=residual_root)
        outs[f"level{lvl}"] = x
    return OrderedDict((k, v) for k, v in outs.items() if k in out_features)


def dla34_forward(sd, x, out_features=("level3", "level4", "level5"), prefix="backbone.bottom_up", hook=None):
    return dla_forward(sd, x, "DLA-34", out_features, prefix, hook)


# ----------------------------------------------------------------------------------------
# [ext] detectron2 FPN + LastLevelP6P7, as built by dla.py:536-561
# ----------------------------------------------------------------------------------------
def fpn_forward(sd, bottom_up, in_features, in_strides, top_block="p6p7", prefix="backbone", hook=None):
    """[ext] detectron2 ``FPN.forward`` (fuse_type "sum", configs/feature_extractors/d2_fpn.yaml:9):
    coarsest first: prev = lateral(C_top); P_top = output(prev); then for each finer level
    prev = lateral(C) + interpolate(prev, x2, nearest); P = output(prev).  The top block is fed the
    *P5 output* because in_feature="p5" is not a bottom-up name (dla.py:550-557); LastLevelP6P7:
    p6 = conv3x3s2(P5), p7 = conv3x3s2(relu(p6)).  Returns OrderedDict p{log2 stride} finest first."""
    stages = [int(math.log2(s)) for s in in_strides]
    results = []
    prev = None

    def norm_of(conv):  # [ext] FPN.__init__: `use_bias = norm == ""`: the norm modules exist iff FE.FPN.NORM is set
        return f"{conv}.norm" if f"{conv}.norm.weight" in sd else None

    for idx in range(len(in_features)):
        name = in_features[-idx - 1]
        st = stages[-idx - 1]
        lat = conv2d(sd, f"{prefix}.fpn_lateral{st}", bottom_up[name], norm=norm_of(f"{prefix}.fpn_lateral{st}"), hook=hook)
        if idx > 0:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat
        out = conv2d(sd, f"{prefix}.fpn_output{st}", prev, padding=1, norm=norm_of(f"{prefix}.fpn_output{st}"), hook=hook)
        results.insert(0, (f"p{st}", out))
    results = OrderedDict(results)
    if top_block:
        top_in = results[f"p{stages[-1]}"]
        p6 = conv2d(sd, f"{prefix}.top_block.p6", top_in, stride=2, padding=1)
        results[f"p{stages[-1] + 1}"] = p6
        if top_block == "p6p7":
            results[f"p{stages[-1] + 2}"] = conv2d(sd, f"{prefix}.top_block.p7", F.relu(p6), stride=2, padding=1)
    return results


# ----------------------------------------------------------------------------------------
# Heads (tridet/modeling/dd3d/fcos2d.py:130-156, fcos3d.py:160-188)
# ----------------------------------------------------------------------------------------
def _tower(sd, p, x, level, num_convs, hook=None):
    """fcos2d.py:71-93 / fcos3d.py:81-101: num_convs x [3x3 conv (no bias) -> norm[level] -> relu];
    norm is a ModuleListDial so call k uses norm k mod L, i.e. level l uses norm.l
    (tridet/layers/normalization.py:30-40)."""
    for i in range(num_convs):
        x = conv2d(sd, f"{p}.{i}", x, padding=1, norm=f"{p}.{i}.norm.{level}", relu=True, hook=hook)
    return x


def fcos2d_head(sd, features, num_convs=4, prefix="fcos2d_head", hook=None, num_box_convs=None):
    """fcos2d.py:130-156 FCOS2DHead.forward (v2; the Scale modules exist in the state dict iff USE_SCALE, fcos2d.py:104-108,146-148)."""
    logits, box2d_reg, centerness, cls_tower_out = [], [], [], []
    for l, f in enumerate(features):
        ct = _tower(sd, prefix + ".cls_tower", f, l, num_convs, hook)
        bt = _tower(sd, prefix + ".box2d_tower", f, l, num_convs if num_box_convs is None else num_box_convs, hook)  # fcos2d.py:54
        logits.append(conv2d(sd, prefix + ".cls_logits", ct, padding=1))
        centerness.append(conv2d(sd, prefix + ".centerness", bt, padding=1))
        reg = conv2d(sd, prefix + ".box2d_reg", bt, padding=1)
        if f"{prefix}.scales_box2d_reg.{l}.scale" in sd:
            reg = reg * sd[f"{prefix}.scales_box2d_reg.{l}.scale"]  # Scale, normalization.py:12-18
        box2d_reg.append(F.relu(reg))
        cls_tower_out.append(ct)
    return logits, box2d_reg, centerness, cls_tower_out


def fcos3d_head(sd, features, num_convs=4, prefix="fcos3d_head", hook=None):
    """fcos3d.py:160-188 FCOS3DHead.forward.  PER_LEVEL_PREDICTORS / USE_SCALE are read off the state dict: one predictor per level
    exists iff PER_LEVEL_PREDICTORS (fcos3d.py:104), the Scale / Offset modules exist iff USE_SCALE (fcos3d.py:128-139)."""
    quat, ctr, depth, size, conf = [], [], [], [], []
    per_level = f"{prefix}.box3d_quat.1.weight" in sd
    use_scale = f"{prefix}.scales_proj_ctr.0.scale" in sd
    for l, f in enumerate(features):
        t = _tower(sd, prefix + ".box3d_tower", f, l, num_convs, hook)
        _l = l if per_level else 0  # fcos3d.py:166
        q = conv2d(sd, f"{prefix}.box3d_quat.{_l}", t, padding=1)
        c = conv2d(sd, f"{prefix}.box3d_ctr.{_l}", t, padding=1)
        d = conv2d(sd, f"{prefix}.box3d_depth.{_l}", t, padding=1)
        s = conv2d(sd, f"{prefix}.box3d_size.{_l}", t, padding=1)
        cf = conv2d(sd, f"{prefix}.box3d_conf.{_l}", t, padding=1)
        if use_scale:  # fcos3d.py:175-180
            c = c * sd[f"{prefix}.scales_proj_ctr.{l}.scale"]
            s = s * sd[f"{prefix}.scales_size.{l}.scale"]
            cf = cf * sd[f"{prefix}.scales_conf.{l}.scale"]
            d = d * sd[f"{prefix}.scales_depth.{l}.scale"] + sd[f"{prefix}.offsets_depth.{l}.bias"]
        quat.append(q), ctr.append(c), depth.append(d), size.append(s), conf.append(cf)
    return quat, ctr, depth, size, conf


# ----------------------------------------------------------------------------------------
# Geometry (tridet/utils/geometry.py, tensor2d.py) and [ext] pytorch3d rotation conversions
# ----------------------------------------------------------------------------------------
def compute_features_locations(h, w, stride, offset="none"):
    """tridet/utils/tensor2d.py:6-25: (x, y) = (j*stride, i*stride), row-major, x fastest."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    loc = torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1)
    if offset == "half":
        loc = loc + stride // 2
    return loc


def quaternion_to_matrix(q):
    """[ext] pytorch3d.transforms.quaternion_to_matrix (real part first): two_s = 2/sum(q^2)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ), -1
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    """[ext] pytorch3d: sqrt(max(0, x)) with zero where x <= 0."""
    ret = torch.zeros_like(x)
    m = x > 0
    ret[m] = torch.sqrt(x[m])
    return ret


def matrix_to_quaternion(matrix):
    """[ext] pytorch3d.transforms.matrix_to_quaternion as shipped in pytorch3d 0.5.0-0.6.x (the
    wheels that exist for py38_cu102_pyt190, docker/Dockerfile:119): four candidates, pick the row
    of argmax(q_abs); the overall sign is NOT canonicalised (w may be negative)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(matrix.shape[:-2] + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1
        )
    )
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0]**2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1]**2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2]**2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3]**2], dim=-1),
        ], dim=-2
    )
    flr = torch.tensor(0.1, dtype=q_abs.dtype)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    best = F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return quat_candidates[best, :].reshape(matrix.shape[:-2] + (4,))


def unproject_points2d(points2d, inv_K):
    """geometry.py:86-112: K^-1 [u, v, 1]^T."""
    pts = F.pad(points2d, (0, 1), value=1.0)  # homogenize_points geometry.py:58-74
    return torch.matmul(inv_K, pts.unsqueeze(-1)).squeeze(-1)


def allocentric_to_egocentric(quat, proj_ctr, inv_K):
    """geometry.py:15-55.  Returns (egocentric_quat, renormalised_flag)."""
    R_obj_to_local = quaternion_to_matrix(quat)
    ray = unproject_points2d(proj_ctr, inv_K)
    z = ray / ray.norm(dim=1, keepdim=True)
    y = z.new_tensor([[0., 1., 0.]]) - z[:, 1:2] * z
    y = y / y.norm(dim=1, keepdim=True)
    x = torch.cross(y, z, dim=1)
    R_local_to_global = torch.stack([x, y, z], dim=-1)
    R_obj_to_global = torch.bmm(R_local_to_global, R_obj_to_local)
    ego = matrix_to_quaternion(R_obj_to_global)
    qn = ego.norm(dim=1, keepdim=True)
    renorm = not torch.allclose(qn, torch.as_tensor(1.), atol=1e-3)  # batch-global trigger, geometry.py:48-53
    if renorm:
        ego = ego / qn.clamp(min=EPS)
    return ego, renorm


def predictions_to_boxes3d(
    quat, proj_ctr, depth, size, locations, inv_K, canon_box_sizes, min_depth, max_depth, focal_factor,
    scale_depth_by_focal_lengths=True, quat_is_allocentric=True, depth_is_distance=False
):
    """fcos3d.py:16-52.  Returns dict(quat, proj_ctr, depth (n,1), size, inv_intrinsics) = the fields of
    ``Boxes3D`` (tridet/structures/boxes3d.py:157-167)."""
    quat = quat / quat.norm(dim=1, keepdim=True).clamp(min=EPS)
    quat = quat / quat.norm(dim=1, keepdim=True)
    if scale_depth_by_focal_lengths:
        pixel_size = torch.norm(torch.stack([inv_K[:, 0, 0], inv_K[:, 1, 1]], dim=-1), dim=-1)
        depth = depth / (pixel_size * focal_factor)
    if depth_is_distance:
        depth = depth / unproject_points2d(locations, inv_K).norm(dim=1).clamp(min=EPS)
    depth = depth.reshape(-1, 1).clamp(min_depth, max_depth)
    proj_ctr = proj_ctr + locations
    if quat_is_allocentric:
        quat, _ = allocentric_to_egocentric(quat, proj_ctr, inv_K)
    size = (size.tanh() + 1.) * canon_box_sizes
    return dict(quat=quat, proj_ctr=proj_ctr, depth=depth, size=size, inv_intrinsics=inv_K)


def boxes3d_tvec(b):
    """boxes3d.py:169-173 Boxes3D.tvec = K^-1 [proj_ctr, 1] * depth."""
    return unproject_points2d(b["proj_ctr"], b["inv_intrinsics"]) * b["depth"]


BOX3D_CORNER_SIGNS = torch.tensor(  # boxes3d.py:12-16 BOX3D_CORNER_MAPPING, transposed to (8,3)
    [[1, 1, 1], [1, -1, 1], [1, -1, -1], [1, 1, -1], [-1, 1, 1], [-1, -1, 1], [-1, -1, -1], [-1, 1, -1]],
    dtype=torch.float32
)


def boxes3d_corners(quat, tvec, size):
    """boxes3d.py:47-64 GenericBoxes3D.corners: corner_i = R(q) (0.5*(l,w,h)*sign_i) + tvec with
    (l,w,h) = size[:, [1,0,2]]  ([ext] pytorch3d Transform3d row-vector convention folded)."""
    R = quaternion_to_matrix(quat)
    lwh = size[:, [1, 0, 2]]
    c = lwh.unsqueeze(1) * (0.5 * BOX3D_CORNER_SIGNS).unsqueeze(0)  # (n,8,3)
    return torch.einsum("nij,nkj->nki", R, c) + tvec.unsqueeze(1)


def boxes3d_vectorize(b):
    """boxes3d.py:142-144: [quat(4) wxyz, tvec(3), size(3) WLH]."""
    return torch.cat([b["quat"], boxes3d_tvec(b), b["size"]], dim=1)


# ----------------------------------------------------------------------------------------
# Inference (fcos2d.py:242-367, fcos3d.py:302-399)
# ----------------------------------------------------------------------------------------
def fcos2d_inference_level(logits, box2d_reg, centerness, locations, cfg):
    """fcos2d.py:270-344 forward_for_single_feature_map.  Returns per-image list of dict + cached indices."""
    inf = cfg["DD3D"]["FCOS2D"]["INFERENCE"]
    N, C = logits.shape[:2]
    scores = logits.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
    box2d_reg = box2d_reg.permute(0, 2, 3, 1).reshape(N, -1, 4)
    centerness = centerness.permute(0, 2, 3, 1).reshape(N, -1).sigmoid()
    if inf["THRESH_WITH_CTR"]:
        scores = scores * centerness[:, :, None]
    candidate_mask = scores > inf["PRE_NMS_THRESH"]
    pre_nms_topk = candidate_mask.reshape(N, -1).sum(1).clamp(max=inf["PRE_NMS_TOPK"])
    if not inf["THRESH_WITH_CTR"]:
        scores = scores * centerness[:, :, None]
    results, info = [], []
    for i in range(N):
        mask_i = candidate_mask[i]
        scores_i = scores[i][mask_i]
        inds = mask_i.nonzero(as_tuple=False)
        fg_inds, class_inds = inds[:, 0], inds[:, 1]
        reg_i = box2d_reg[i][fg_inds]
        loc_i = locations[fg_inds]
        k = int(pre_nms_topk[i])
        cls_i = class_inds
        topk_indices = None
        if int(mask_i.sum()) > k:
            scores_i, topk_indices = scores_i.topk(k, sorted=False)
            cls_i, reg_i, loc_i = class_inds[topk_indices], reg_i[topk_indices], loc_i[topk_indices]
        boxes = torch.stack(
            [loc_i[:, 0] - reg_i[:, 0], loc_i[:, 1] - reg_i[:, 1], loc_i[:, 0] + reg_i[:, 2], loc_i[:, 1] + reg_i[:, 3]],
            dim=1
        )
        results.append(dict(pred_boxes=boxes, scores=torch.sqrt(scores_i), pred_classes=cls_i, locations=loc_i))
        info.append(dict(fg_inds=fg_inds, class_inds=class_inds, topk_indices=topk_indices))
    return results, info


def fcos3d_inference_level(quat, ctr, depth, size, conf, inv_intrinsics, instances, info, cfg):
    """fcos3d.py:328-399 forward_for_single_feature_map (class-aware); adds pred_boxes3d / scores_3d in place."""
    c3 = cfg["DD3D"]["FCOS3D"]
    N = quat.shape[0]
    C = cfg["DD3D"]["NUM_CLASSES"] if not c3["CLASS_AGNOSTIC_BOX3D"] else 1
    quat = quat.permute(0, 2, 3, 1).reshape(N, -1, 4, C)
    ctr = ctr.permute(0, 2, 3, 1).reshape(N, -1, 2, C)
    depth = depth.permute(0, 2, 3, 1).reshape(N, -1, C)
    size = size.permute(0, 2, 3, 1).reshape(N, -1, 3, C)
    conf = conf.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
    canon = torch.tensor(c3["CANONICAL_BOX3D_SIZES"], dtype=torch.float32)
    for i in range(N):
        fg, cls, topk = info[i]["fg_inds"], info[i]["class_inds"], info[i]["topk_indices"]
        q_i, c_i, d_i, s_i, cf_i = quat[i][fg], ctr[i][fg], depth[i][fg], size[i][fg], conf[i][fg]
        if c3["CLASS_AGNOSTIC_BOX3D"]:
            q_i, c_i, d_i, s_i, cf_i = q_i.squeeze(-1), c_i.squeeze(-1), d_i.squeeze(-1), s_i.squeeze(-1), cf_i.squeeze(-1)
        else:
            I = cls[..., None, None]
            q_i = torch.gather(q_i, 2, I.repeat(1, 4, 1)).squeeze(-1)
            c_i = torch.gather(c_i, 2, I.repeat(1, 2, 1)).squeeze(-1)
            d_i = torch.gather(d_i, 1, I.squeeze(-1)).squeeze(-1)
            s_i = torch.gather(s_i, 2, I.repeat(1, 3, 1)).squeeze(-1)
            cf_i = torch.gather(cf_i, 1, I.squeeze(-1)).squeeze(-1)
        if topk is not None:
            q_i, c_i, d_i, s_i, cf_i = q_i[topk], c_i[topk], d_i[topk], s_i[topk], cf_i[topk]
        inst = instances[i]
        inst["scores_3d"] = inst["scores"] * cf_i
        canon_i = canon[inst["pred_classes"]]
        inv_K = inv_intrinsics[i][None].expand(len(q_i), 3, 3)
        inst["pred_boxes3d"] = predictions_to_boxes3d(
            q_i, c_i, d_i, s_i, inst["locations"], inv_K, canon_i, c3["MIN_DEPTH"], c3["MAX_DEPTH"],
            c3["SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR"], c3["SCALE_DEPTH_BY_FOCAL_LENGTHS"],
            c3["PREDICT_ALLOCENTRIC_ROT"], c3["PREDICT_DISTANCE"]
        )


# ----------------------------------------------------------------------------------------
# [ext] torchvision 0.10.0 nms / batched_nms (through detectron2.layers.batched_nms)
# ----------------------------------------------------------------------------------------
def nms(boxes, scores, iou_threshold):
    """[ext] torchvision.ops.nms: sort by score desc (stable here; the reference's tie order is
    implementation-defined), greedy; suppress j when inter/(area_i+area_j-inter) > thr; no +1.
    Returns int64 indices of kept boxes in descending-score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0, ), dtype=torch.int64)
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = torch.sort(scores, descending=True, stable=True)[1]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for _i in range(n):
        i = int(order[_i])
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.numel() == 0:
            break
        xx1 = torch.maximum(x1[i], x1[rest])
        yy1 = torch.maximum(y1[i], y1[rest])
        xx2 = torch.minimum(x2[i], x2[rest])
        yy2 = torch.minimum(y2[i], y2[rest])
        w = (xx2 - xx1).clamp(min=0)
        h = (yy2 - yy1).clamp(min=0)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > iou_threshold]] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """[ext] detectron2.layers.batched_nms (<40000 boxes) -> torchvision 0.10 batched_nms(boxes.float(), ...):
    ``boxes.numel() > 4000`` -> per-class loop (_batched_nms_vanilla), else coordinate trick
    (boxes + idxs*(boxes.max()+1))."""
    boxes = boxes.float()
    if boxes.numel() == 0:
        return torch.empty((0, ), dtype=torch.int64)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for class_id in torch.unique(idxs):
            curr = torch.where(idxs == class_id)[0]
            keep_mask[curr[nms(boxes[curr], scores[curr], iou_threshold)]] = True
        keep_indices = torch.where(keep_mask)[0]
        return keep_indices[torch.sort(scores[keep_indices], descending=True, stable=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def _index_instances(inst, idx):
    """[ext] detectron2 Instances.__getitem__: index every field (Boxes3D fields via boxes3d.py:248-270)."""
    out = {}
    for k, v in inst.items():
        out[k] = {kk: vv[idx] for kk, vv in v.items()} if isinstance(v, dict) else v[idx]
    return out


def _cat_instances(insts):
    """[ext] detectron2 Instances.cat + Boxes3D.cat (boxes3d.py:221-236)."""
    out = {}
    for k, v in insts[0].items():
        if isinstance(v, dict):
            out[k] = {kk: torch.cat([x[k][kk] for x in insts], 0) for kk in v}
        else:
            out[k] = torch.cat([x[k] for x in insts], 0)
    return out


def nms_and_top_k(inst, cfg, score_key):
    """fcos2d.py:346-367: batched NMS ranked by ``score_key`` then top-k thresholding on 2D ``scores`` (>=)."""
    inf = cfg["DD3D"]["FCOS2D"]["INFERENCE"]
    if inf["NMS_THRESH"] > 0:
        keep = batched_nms(inst["pred_boxes"], inst[score_key], inst["pred_classes"], inf["NMS_THRESH"])
        inst = _index_instances(inst, keep)
    n = len(inst["scores"])
    if n > inf["POST_NMS_TOPK"] > 0:
        scores = inst["scores"]
        thr, _ = torch.kthvalue(scores, n - inf["POST_NMS_TOPK"] + 1)
        keep = torch.nonzero(scores >= thr.item()).squeeze(1)
        inst = _index_instances(inst, keep)
    return inst


def detector_postprocess(inst, image_size, out_h, out_w):
    """[ext] detectron2.modeling.postprocessing.detector_postprocess: scale 2D boxes by
    (out_w/in_w, out_h/in_h), clip to the output size, drop empty boxes (filters every field)."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    boxes = inst["pred_boxes"].clone()
    boxes[:, 0::2] *= sx
    boxes[:, 1::2] *= sy
    boxes[:, 0].clamp_(min=0, max=out_w)
    boxes[:, 1].clamp_(min=0, max=out_h)
    boxes[:, 2].clamp_(min=0, max=out_w)
    boxes[:, 3].clamp_(min=0, max=out_h)
    inst = dict(inst)
    inst["pred_boxes"] = boxes
    keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    return _index_instances(inst, keep)


# ----------------------------------------------------------------------------------------
# Pre-processing (core.py:61-72, image_list.py:94-158)
# ----------------------------------------------------------------------------------------
def preprocess(sd, batched_inputs, size_divisibility):
    """core.py:65-72: (x - pixel_mean)/pixel_std per image, THEN pad right/bottom with 0.0 to a multiple of
    ``size_divisibility`` (ImageList.from_tensors, image_list.py:120-142)."""
    images = [(x["image"].to(torch.float32) - sd["pixel_mean"]) / sd["pixel_std"] for x in batched_inputs]
    image_sizes = [(im.shape[-2], im.shape[-1]) for im in images]
    H = max(s[0] for s in image_sizes)
    W = max(s[1] for s in image_sizes)
    if size_divisibility > 1:
        H = (H + size_divisibility - 1) // size_divisibility * size_divisibility
        W = (W + size_divisibility - 1) // size_divisibility * size_divisibility
    batch = images[0].new_zeros((len(images), images[0].shape[0], H, W))
    for im, dst in zip(images, batch):
        dst[:, :im.shape[-2], :im.shape[-1]].copy_(im)
    intrinsics = torch.stack([x["intrinsics"].to(torch.float32) for x in batched_inputs], 0)
    if torch.allclose(intrinsics[0], torch.eye(3)):
        raise ValueError("Intrinsics is Identity.")  # image_list.py:57-62
    return batch, image_sizes, intrinsics


# ----------------------------------------------------------------------------------------
# DD3D.forward, inference branch (core.py:64-164)
# ----------------------------------------------------------------------------------------
def dd3d_backbone(sd, cfg, x, hook=None):
    """feature_extractor/__init__.py:13-26 -> cfg.FE.BUILDER.  Returns (features list finest-first, strides)."""
    builder = cfg["FE"]["BUILDER"]
    if builder == "build_fcos_dla_fpn_backbone_p67":  # dla.py:536-561
        feats = cfg["FE"]["BACKBONE"]["OUT_FEATURES"]
        strides = [2**int(f[len("level"):]) for f in feats]
        bu = dla_forward(sd, x, cfg["FE"]["BACKBONE"]["NAME"], tuple(feats), hook=hook)
        fpn = fpn_forward(sd, bu, feats, strides, top_block="p6p7", hook=hook)
    elif builder == "build_fcos_vovnet_fpn_backbone_p6":
        from oracle.vovnet_oracle import vovnet_forward  # vovnet.py:428-454
        feats = cfg["FE"]["BACKBONE"]["OUT_FEATURES"]
        strides = [2**int(f[len("stage"):]) for f in feats]
        bu = vovnet_forward(sd, x, cfg["FE"]["BACKBONE"]["NAME"], tuple(feats), hook=hook)
        fpn = fpn_forward(sd, bu, feats, strides, top_block="p6", hook=hook)
    else:
        raise KeyError(builder)
    names = list(cfg["DD3D"]["IN_FEATURES"] or fpn.keys())  # core.py:32-34,84: the heads see DD3D.IN_FEATURES (default: every output)
    return [fpn[n] for n in names], [2**int(n[1:]) for n in names], bu


def size_divisibility(cfg):
    """[ext] FPN._size_divisibility = stride of the coarsest bottom-up feature (32) x4 (dla.py:559) / x2 (vovnet.py:452)."""
    return {"build_fcos_dla_fpn_backbone_p67": 128, "build_fcos_vovnet_fpn_backbone_p6": 64}[cfg["FE"]["BUILDER"]]


def dd3d_forward(sd, cfg, batched_inputs, hook=None, stop_after_heads=False):
    """core.py:64-164 DD3D.forward with self.training == False.  Returns (results, stages):
    results = list (per image) of dict with the ``Instances`` fields; stages = intermediate tensors."""
    stages = {}
    x, image_sizes, intrinsics = preprocess(sd, batched_inputs, size_divisibility(cfg))
    stages["images"] = x
    features, strides, bu = dd3d_backbone(sd, cfg, x, hook)
    stages["bottom_up"] = bu
    stages["features"] = features
    locations = [
        compute_features_locations(f.shape[-2], f.shape[-1], s, cfg["DD3D"]["FEATURE_LOCATIONS_OFFSET"])
        for f, s in zip(features, strides)
    ]
    logits, box2d_reg, centerness, cls_tower_out = fcos2d_head(
        sd, features, cfg["DD3D"]["FCOS2D"]["NUM_CLS_CONVS"], hook=hook, num_box_convs=cfg["DD3D"]["FCOS2D"]["NUM_BOX_CONVS"]
    )
    stages.update(logits=logits, box2d_reg=box2d_reg, centerness=centerness)
    if cfg["MODEL"]["BOX3D_ON"]:  # core.py:38-42,90-92: without it the model is `only_box2d`
        quat, ctr, depth, size, conf = fcos3d_head(sd, features, cfg["DD3D"]["FCOS3D"]["NUM_CONVS"], hook=hook)
        stages.update(quat=quat, ctr=ctr, depth=depth, size=size, conf=conf)
    if stop_after_heads:
        return None, stages
    inv_intrinsics = intrinsics.inverse()  # core.py:93 (the reference's ImageList.intrinsics fails without 'intrinsics' in the inputs)
    stages["inv_intrinsics"] = inv_intrinsics
    if cfg["DD3D"]["INFERENCE"]["DO_BEV_NMS"]:  # core.py:135-150 lives with the other BEV code
        from oracle.nuscenes_oracle import nuscenes_postprocess_from_heads
        results, stages2 = nuscenes_postprocess_from_heads(
            cfg, stages, locations, inv_intrinsics, image_sizes, batched_inputs, sample_aggregate=False
        )
    else:
        results, stages2 = dd3d_postprocess_from_heads(cfg, stages, locations, inv_intrinsics, image_sizes, batched_inputs)
    stages.update(stages2)
    return results, stages


def dd3d_postprocess_from_heads(cfg, heads, locations, inv_intrinsics, image_sizes, batched_inputs):
    """core.py:114-164 from the head maps on: 2D inference, 3D decode, cat levels, NMS/top-k, resize."""
    L = len(heads["logits"])
    pred = []  # (L, B)
    infos = []
    for l in range(L):
        r, info = fcos2d_inference_level(heads["logits"][l], heads["box2d_reg"][l], heads["centerness"][l], locations[l], cfg)
        for inst in r:
            inst["fpn_levels"] = torch.ones(len(inst["scores"]), dtype=torch.long) * l  # fcos2d.py:263-264
        if "quat" in heads:  # core.py:117-125 (not only_box2d)
            fcos3d_inference_level(
                heads["quat"][l], heads["ctr"][l], heads["depth"][l], heads["size"][l], heads["conf"][l], inv_intrinsics, r,
                info, cfg
            )
        pred.append(r)
        infos.append(info)
    B = len(image_sizes)
    per_image = [_cat_instances([pred[l][i] for l in range(L)]) for i in range(B)]  # core.py:130-131
    stages = {"candidates": per_image, "level_info": infos}
    inf = cfg["DD3D"]["INFERENCE"]
    if inf["DO_NMS"]:
        score_key = "scores_3d" if "quat" in heads else "scores"  # core.py:125-127
        per_image = [nms_and_top_k(x, cfg, score_key) for x in per_image]  # core.py:134-135
    stages["after_nms"] = per_image
    results = []
    for inst, inp, isz in zip(per_image, batched_inputs, image_sizes):
        if inf["DO_POSTPROCESS"]:
            inst = detector_postprocess(inst, isz, inp.get("height", isz[0]), inp.get("width", isz[1]))
        results.append(inst)
    return results, stages
