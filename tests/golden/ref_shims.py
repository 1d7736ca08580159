"""Shims that let the REAL reference modules (/root/reference/tridet/...) be imported in this image.

detectron2 / fvcore / pytorch3d / pyquaternion / mpi4py / cv2 are not installed (and there is no network).  The
reference's forward path only needs a thin slice of them; this file installs stand-in modules for exactly that slice
into ``sys.modules`` so that ``tridet.modeling.dd3d.core.DD3D`` -- the reference's own code -- can be constructed
and run on CPU to produce golden vectors (tests/golden/make_golden.py).

What that pins and what it does not:
  * pinned: every line of the in-repo path (core.py, fcos2d.py, fcos3d.py, dla.py, normalization.py, boxes3d.py,
    image_list.py, geometry.py, tensor2d.py) runs as written by the reference authors;
  * NOT pinned: the third-party pieces below are re-statements of the published behaviour of detectron2 v0.5/0.6,
    torchvision 0.10 and pytorch3d 0.5/0.6 -- "[ext] parity unpinned".  They are written HERE, independently
    of the oracle and of the package under test (nothing in this file imports `oracle` or `dd3d_amd`): NMS is a brute-force O(n^2)
    greedy loop, rotated IoU a float64 polygon clipper, matrix -> quaternion is scipy's, Boxes / Instances are local classes -- so the
    goldens cross-check the oracle's restatements instead of echoing them.

Test infrastructure only; never imported by dd3d_amd.
"""
import functools
import math
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_ROOT = os.environ.get("DD3D_REFERENCE_ROOT", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


# --------------------------------------------------------------------------------------------- detectron2.layers [ext]
class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, training=False, eps=self.eps)


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    return {"BN": nn.BatchNorm2d, "FrozenBN": FrozenBatchNorm2d}[norm](out_channels)


class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class ShapeSpec(tuple):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        self = super().__new__(cls, (channels, height, width, stride))
        return self

    channels = property(lambda s: s[0])
    height = property(lambda s: s[1])
    width = property(lambda s: s[2])
    stride = property(lambda s: s[3])


def cat(tensors, dim=0):
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def _nms(boxes, scores, thr):
    """[ext] torchvision.ops.nms (0.10), written HERE as a brute-force O(n^2) greedy loop -- deliberately not the
    oracle's (or the product's) implementation, so that the goldens cross-check those: visit the boxes by descending score, keep a box
    unless a kept one overlaps it with IoU = inter / (area_a + area_b - inter) > thr (float32, no +1, clamped overlaps); returns the
    kept indices in visiting order."""
    b = boxes.detach().to(torch.float32).cpu().numpy()
    order = torch.sort(scores.detach().float().cpu(), descending=True, stable=True)[1].tolist()
    import numpy as np
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    kept = []
    for i in order:
        ok = True
        for j in kept:
            w = np.float32(max(np.float32(0.0), np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]))))
            h = np.float32(max(np.float32(0.0), np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]))))
            inter = np.float32(w * h)
            if np.float32(inter / np.float32(np.float32(area[i] + area[j]) - inter)) > np.float32(thr):
                ok = False
                break
        if ok:
            kept.append(i)
    return torch.tensor(kept, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """[ext] detectron2.layers.batched_nms -> torchvision.ops.batched_nms (0.10): float boxes; more than 4000 coordinates -> one NMS per
    class, survivors re-sorted by score; else the coordinate trick (every class shifted by max coordinate + 1, one NMS)."""
    boxes = boxes.float()
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0, ), dtype=torch.int64)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for class_id in torch.unique(idxs):
            curr = torch.where(idxs == class_id)[0]
            keep_mask[curr[_nms(boxes[curr], scores[curr], iou_threshold)]] = True
        keep = torch.where(keep_mask)[0]
        return keep[scores[keep].sort(descending=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return _nms(boxes + offsets[:, None], scores, iou_threshold)


# ---- [ext] detectron2.structures.Boxes / Instances (appendix A7), written here rather than borrowed from the package under test
class Boxes:
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        self.tensor = torch.stack((self.tensor[:, 0].clamp(min=0, max=w), self.tensor[:, 1].clamp(min=0, max=h),
                                   self.tensor[:, 2].clamp(min=0, max=w), self.tensor[:, 3].clamp(min=0, max=h)), dim=-1)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device


class Instances:
    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields = {}
        for k, v in kwargs.items():
            self.set(k, v)

    image_size = property(lambda s: s._image_size)

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        if len(self._fields):
            assert len(self) == len(value), f"Adding a field of length {len(value)} to a Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *args, **kwargs):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        if isinstance(item, int):
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(instance_lists):
        assert all(isinstance(i, Instances) for i in instance_lists) and len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        for i in instance_lists[1:]:
            assert i.image_size == image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = [x for v in values for x in v]
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError(f"Unsupported type {type(v0)} for concatenation")
            ret.set(k, values)
        return ret


def quaternion_to_matrix(quaternions):
    """[ext] pytorch3d.transforms.quaternion_to_matrix (appendix A8), real part first."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r), two_s * (i * j + k * r), 1 - two_s * (i * i + k * k),
                     two_s * (j * k - i * r), two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix):
    """[ext] pytorch3d.transforms.matrix_to_quaternion: NOT the oracle's restatement of pytorch3d's branches but scipy's independent
    conversion (float64, Shepperd's method), returned real part first in float32.  The overall sign differs between pytorch3d releases
    (appendix A8) -- every consumer compares quaternions up to sign or through the rotation / the box corners."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    m = matrix.detach().cpu().double().numpy().reshape(-1, 3, 3)
    if m.shape[0] == 0:
        return matrix.new_zeros(matrix.shape[:-2] + (4, ))
    q = Rotation.from_matrix(m).as_quat()  # (x, y, z, w)
    q = np.concatenate([q[:, 3:4], q[:, 0:3]], axis=1)
    return torch.from_numpy(q).to(matrix.dtype).reshape(matrix.shape[:-2] + (4, )).to(matrix.device)


# --------------------------------------------------------------------------------------------- registries / config
class Registry:
    def __init__(self, name):
        self._name, self._obj_map = name, {}

    def register(self, obj=None):
        if obj is None:

            def deco(f):
                self._obj_map[f.__name__] = f
                return f

            return deco
        self._obj_map[obj.__name__] = obj

    def get(self, name):
        return self._obj_map[name]


def configurable(init_func):
    """[ext] detectron2.config.configurable for __init__: called with a cfg first -> cls.from_config(cfg, ...)."""
    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        first = args[0] if args else kwargs.get("cfg")
        if first is not None and hasattr(first, "keys") and hasattr(type(self), "from_config"):
            # detectron2 _get_args_from_config: keyword arguments from_config does not name go to __init__ unchanged
            import inspect
            params = inspect.signature(type(self).from_config).parameters
            if any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in params.values()):
                explicit = type(self).from_config(*args, **kwargs)
            else:
                extra = {k: kwargs.pop(k) for k in list(kwargs) if k not in params}
                explicit = type(self).from_config(*args, **kwargs)
                explicit.update(extra)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)

    return wrapped


# --------------------------------------------------------------------------------------------- backbone / FPN [ext]
class Backbone(nn.Module):
    def output_shape(self):
        return {
            name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
            for name in self._out_features
        }

    @property
    def size_divisibility(self):
        return 0


class LastLevelMaxPool(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_levels, self.in_feature = 1, "p5"

    def forward(self, x):
        return [F.max_pool2d(x, kernel_size=1, stride=2, padding=0)]


def _xavier(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class LastLevelP6P7(nn.Module):
    def __init__(self, in_channels, out_channels, in_feature="res5"):
        super().__init__()
        self.num_levels, self.in_feature = 2, in_feature
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, 2, 1)
        _xavier(self.p6), _xavier(self.p7)

    def forward(self, c5):
        p6 = self.p6(c5)
        p7 = self.p7(F.relu(p6))
        return [p6, p7]


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels_per_feature = [input_shapes[f].channels for f in in_features]
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(in_channels_per_feature):
            lateral_conv = Conv2d(in_channels, out_channels, kernel_size=1, bias=use_bias, norm=get_norm(norm, out_channels))
            output_conv = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, out_channels))
            _xavier(lateral_conv), _xavier(output_conv)
            stage = int(math.log2(strides[idx]))
            self.add_module("fpn_lateral{}".format(stage), lateral_conv)
            self.add_module("fpn_output{}".format(stage), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in strides}
        if self.top_block is not None:
            for s in range(stage, stage + self.top_block.num_levels):
                self._out_feature_strides["p{}".format(s + 1)] = 2**(s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._fuse_type = fuse_type

    # plain python lists of registered modules must not be registered twice
    def __setattr__(self, k, v):
        if k in ("lateral_convs", "output_convs"):
            object.__setattr__(self, k, v)
        else:
            super().__setattr__(k, v)

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        bottom_up_features = self.bottom_up(x)
        results = []
        prev_features = self.lateral_convs[0](bottom_up_features[self.in_features[-1]])
        results.append(self.output_convs[0](prev_features))
        for idx, (lateral_conv, output_conv) in enumerate(zip(self.lateral_convs, self.output_convs)):
            if idx > 0:
                features = bottom_up_features[self.in_features[-idx - 1]]
                top_down_features = F.interpolate(prev_features, scale_factor=2.0, mode="nearest")
                lateral_features = lateral_conv(features)
                prev_features = lateral_features + top_down_features
                if self._fuse_type == "avg":
                    prev_features /= 2
                results.insert(0, output_conv(prev_features))
        if self.top_block is not None:
            if self.top_block.in_feature in bottom_up_features:
                top_block_in_feature = bottom_up_features[self.top_block.in_feature]
            else:
                top_block_in_feature = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(top_block_in_feature))
        assert len(self._out_features) == len(results)
        return {f: res for f, res in zip(self._out_features, results)}


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    boxes = results.pred_boxes
    boxes.tensor[:, 0::2] *= scale_x
    boxes.tensor[:, 1::2] *= scale_y
    h, w = results.image_size
    boxes.tensor[:, 0].clamp_(min=0, max=w)
    boxes.tensor[:, 1].clamp_(min=0, max=h)
    boxes.tensor[:, 2].clamp_(min=0, max=w)
    boxes.tensor[:, 3].clamp_(min=0, max=h)
    return results[boxes.nonempty()]


# --------------------------------------------------------------------------------------------- pytorch3d [ext]
class Transform3d:
    """[ext] pytorch3d.transforms.Transform3d, the slice postprocessing.py / bev_nms.py / boxes3d.py use: row-vector
    convention (points @ M), 4x4 matrices batched on dim 0, ``compose`` = apply self first, then the others."""
    def __init__(self, dtype=torch.float32, device="cpu", matrix=None):
        if matrix is None:
            self._matrix = torch.eye(4, dtype=dtype, device=device).view(1, 4, 4)
        else:
            self._matrix = matrix.view(-1, 4, 4)
        self._others = []

    def compose(self, *others):
        out = Transform3d(matrix=self._matrix)
        out._others = self._others + list(others)
        return out

    def get_matrix(self):
        m = self._matrix
        for o in self._others:
            m = torch.matmul(m, o.get_matrix())  # broadcast over the batch
        return m

    def transform_points(self, points):
        p = points if points.dim() == 3 else points[None]
        ones = torch.ones(p.shape[0], p.shape[1], 1, dtype=p.dtype, device=p.device)
        out = torch.bmm(torch.cat([p, ones], dim=2), self.get_matrix().expand(p.shape[0], 4, 4))
        out = out[..., :3] / out[..., 3:]
        return out if points.dim() == 3 else out[0]


class Translate(Transform3d):
    def __init__(self, x, y=None, z=None, dtype=torch.float32, device="cpu"):
        xyz = x if y is None else torch.stack([torch.as_tensor(x), torch.as_tensor(y), torch.as_tensor(z)], -1)
        xyz = xyz.view(-1, 3)
        m = torch.eye(4, dtype=xyz.dtype, device=xyz.device).view(1, 4, 4).repeat(xyz.shape[0], 1, 1)
        m[:, 3, :3] = xyz
        super().__init__(matrix=m)


class Rotate(Transform3d):
    def __init__(self, R, dtype=torch.float32, device="cpu", orthogonal_tol=1e-5):
        R = R.view(-1, 3, 3)
        m = torch.eye(4, dtype=R.dtype, device=R.device).view(1, 4, 4).repeat(R.shape[0], 1, 1)
        m[:, :3, :3] = R
        super().__init__(matrix=m)


class RotatedBoxes:
    """[ext] detectron2.structures.RotatedBoxes: (cx, cy, w, h, angle_deg CCW) rows."""
    def __init__(self, tensor):
        self.tensor = tensor.reshape(-1, 5).to(torch.float32)

    def __len__(self):
        return self.tensor.shape[0]


def _poly_clip_area(b1, b2):
    """Intersection area of two rotated rectangles by Sutherland-Hodgman clipping in float64 -- deliberately NOT the
    detectron2 algorithm the oracle restates, so that the goldens cross-check it."""
    import math

    def verts(b):
        cx, cy, w, h, a = [float(v) for v in b]
        th = math.radians(a)
        c, s = math.cos(th), math.sin(th)
        # detectron2 convention (box_iou_rotated_utils.h get_rotated_vertices): vertex = centre + R(theta) (dx, dy)
        return [(cx + c * dx - s * dy, cy + s * dx + c * dy) for dx, dy in ((-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2))]

    def area(p):
        return 0.5 * sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p)))

    subj, clip = verts(b1), verts(b2)
    if area(clip) < 0:
        clip = clip[::-1]
    for i in range(4):
        a, b = clip[i], clip[(i + 1) % 4]
        inside = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0
        out = []
        for j in range(len(subj)):
            p, q = subj[j], subj[(j + 1) % len(subj)]
            ip, iq = inside(p), inside(q)
            if ip != iq:
                d1 = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
                d2 = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
                t = d1 / (d1 - d2)
                x = (p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]))
                if ip:
                    out += [x]
                else:
                    out += [x, q]
            elif ip:
                out.append(q)
        subj = out
        if len(subj) < 3:
            return 0.0
    return abs(area(subj))


def _nms_rotated(boxes, scores, thr):
    order = torch.argsort(scores, descending=True, stable=True).tolist()
    b = boxes.double().tolist()
    keep, dead = [], set()
    for ii, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        ai = b[i][2] * b[i][3]
        for j in order[ii + 1:]:
            if j in dead:
                continue
            aj = b[j][2] * b[j][3]
            if max(abs(b[i][0] - b[j][0]), abs(b[i][1] - b[j][1])) > 0.5 * (math.hypot(b[i][2], b[i][3]) + math.hypot(b[j][2], b[j][3])):
                continue
            inter = _poly_clip_area(b[i], b[j])
            if inter / max(ai + aj - inter, 1e-30) > thr:
                dead.add(j)
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms_rotated(boxes, scores, idxs, iou_threshold):
    """[ext] detectron2.layers.nms.batched_nms_rotated: offset the centres per category, one nms_rotated call."""
    assert boxes.shape[-1] == 5
    if boxes.numel() == 0:
        return torch.empty((0, ), dtype=torch.int64)
    boxes = boxes.float()
    max_coordinate = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    min_coordinate = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.max(boxes[:, 2], boxes[:, 3]) / 2).min()
    offsets = idxs.to(boxes) * (max_coordinate - min_coordinate + 1)
    b = boxes.clone()
    b[:, :2] += offsets[:, None]
    return _nms_rotated(b, scores, iou_threshold)


class Quaternion:
    """[ext] the slice of pyquaternion.Quaternion that tridet/structures/pose.py uses (w, x, y, z; numpy float64)."""
    def __init__(self, *args, **kwargs):
        import numpy as np
        if "matrix" in kwargs:
            m = np.asarray(kwargs["matrix"], dtype=np.float64)[:3, :3]
            # pyquaternion._from_matrix: trace method (Shepperd)
            t = np.trace(m)
            if t > 0:
                s = 0.5 / np.sqrt(t + 1.0)
                q = [0.25 / s, (m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s]
            elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
                s = 2.0 * np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
                q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
            elif m[1, 1] > m[2, 2]:
                s = 2.0 * np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2])
                q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
            else:
                s = 2.0 * np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1])
                q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
            self.q = np.asarray(q, dtype=np.float64)
        elif "axis" in kwargs:
            # pyquaternion._from_axis_angle: axis normalised unless |1 - |axis|^2| <= 1e-12; q = (cos(a/2), axis sin(a/2))
            import math
            axis = np.asarray(kwargs["axis"], dtype=np.float64)
            angle = kwargs["radians"] if "radians" in kwargs else kwargs.get("angle", np.deg2rad(kwargs.get("degrees", 0.0)))
            mag_sq = float(axis @ axis)
            if abs(1.0 - mag_sq) > 1e-12:
                axis = axis / math.sqrt(mag_sq)
            self.q = np.concatenate([[math.cos(angle / 2.0)], axis * math.sin(angle / 2.0)])
        elif len(args) == 1 and isinstance(args[0], Quaternion):
            self.q = args[0].q.copy()
        elif len(args) == 1:
            self.q = np.asarray(args[0], dtype=np.float64).reshape(4).copy()
        elif len(args) == 4:
            self.q = np.asarray(args, dtype=np.float64)
        else:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])

    elements = property(lambda s: s.q)

    def __mul__(self, o):
        import numpy as np
        w1, x1, y1, z1 = self.q
        w2, x2, y2, z2 = o.q
        return Quaternion(np.array([
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
        ]))

    @property
    def inverse(self):
        import numpy as np
        return Quaternion(self.q * np.array([1.0, -1.0, -1.0, -1.0]) / float(self.q @ self.q))

    def _normalise(self):
        """pyquaternion: in place, only when the quaternion is not unit within 1e-14."""
        import math
        ss = float(self.q @ self.q)
        if not abs(1.0 - ss) < 1e-14 and ss > 0:
            self.q = self.q / math.sqrt(ss)

    @property
    def axis(self):
        import numpy as np
        self._normalise()
        n = np.linalg.norm(self.q[1:])
        return np.zeros(3) if n < 1e-17 else self.q[1:] / n

    @property
    def angle(self):
        import math
        import numpy as np
        self._normalise()
        theta = 2.0 * math.atan2(np.linalg.norm(self.q[1:]), self.q[0])
        r = ((theta + math.pi) % (2 * math.pi)) - math.pi  # _wrap_angle
        return math.pi if r == -math.pi else r

    @property
    def rotation_matrix(self):
        import numpy as np
        w, x, y, z = self.q / np.linalg.norm(self.q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    @property
    def transformation_matrix(self):
        import numpy as np
        t = np.eye(4)
        t[:3, :3] = self.rotation_matrix
        return t

    def rotate(self, v):
        import numpy as np
        return self.rotation_matrix @ np.asarray(v, dtype=np.float64)



# --------------------------------------------------------------------------------------------- fvcore / detectron2 transforms [ext]
class Transform:
    """[ext] fvcore.transforms.transform.Transform: apply_image / apply_coords, apply_box via the four corners, extra data types
    registered per class (register_type -> apply_<type>)."""
    @classmethod
    def register_type(cls, data_type, func):
        setattr(cls, "apply_" + data_type, lambda self, x, _f=func: _f(self, x))

    def apply_box(self, box):
        import numpy as np
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)

    def inverse(self):
        raise NotImplementedError


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return self

    def __getattr__(self, name):
        if name.startswith("apply_"):
            return lambda x: x
        raise AttributeError(name)


class HFlipTransform(Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        import numpy as np
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords

    def inverse(self):
        return self


class VFlipTransform(Transform):
    def __init__(self, height):
        self.height = height

    def inverse(self):
        return self


class ResizeTransform(Transform):
    """[ext] detectron2.data.transforms.ResizeTransform: uint8 images through PIL (the real library is installed)."""
    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w, self.interp = h, w, new_h, new_w, interp

    def apply_image(self, img, interp=None):
        import numpy as np
        from PIL import Image
        assert img.shape[:2] == (self.h, self.w) and img.dtype == np.uint8
        return np.asarray(Image.fromarray(img).resize((self.new_w, self.new_h), Image.BILINEAR))

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def inverse(self):
        return ResizeTransform(self.new_h, self.new_w, self.h, self.w, self.interp)


class TransformList(Transform):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __getattr__(self, name):
        if name.startswith("apply_"):
            def run(x, _n=name):
                for t in self.transforms:
                    x = getattr(t, _n)(x)
                return x
            return run
        raise AttributeError(name)

    def __add__(self, other):
        others = other.transforms if isinstance(other, TransformList) else [other]
        return TransformList(self.transforms + others)

    def __radd__(self, other):
        others = other.transforms if isinstance(other, TransformList) else [other]
        return TransformList(others + self.transforms)

    def inverse(self):
        return TransformList([t.inverse() for t in self.transforms[::-1]])


class D2ResizeShortestEdge:
    """[ext] detectron2 ResizeShortestEdge with an int short_edge_length ("range" over (s, s))."""
    def __init__(self, short_edge_length, max_size=sys.maxsize, sample_style="range", interp=None):
        self.short_edge_length, self.max_size = short_edge_length, max_size

    def get_transform(self, image):
        h, w = image.shape[:2]
        size = self.short_edge_length
        scale = size * 1.0 / min(h, w)
        newh, neww = (size, scale * w) if h < w else (scale * h, size)
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return ResizeTransform(h, w, int(newh + 0.5), int(neww + 0.5))


class D2RandomFlip:
    def __init__(self, prob=0.5, *, horizontal=True, vertical=False):
        self.prob, self.horizontal = prob, horizontal

    def get_transform(self, image):
        h, w = image.shape[:2]
        assert self.prob >= 1.0 and self.horizontal
        return HFlipTransform(w)


def apply_augmentations(augmentations, image):
    tfms = []
    for aug in augmentations:
        t = aug.get_transform(image)
        image = t.apply_image(image)
        tfms.append(t)
    return image, TransformList(tfms)


def install():
    """Install every shim module and make ``tridet`` importable without running its package __init__ chains."""
    if "detectron2" in sys.modules and getattr(sys.modules["detectron2"], "_dd3d_shim", False):
        return
    d2 = _mod("detectron2", _dd3d_shim=True)
    _mod("detectron2.config", configurable=configurable)
    _mod("detectron2.layers", Conv2d=Conv2d, get_norm=get_norm, FrozenBatchNorm2d=FrozenBatchNorm2d, ShapeSpec=ShapeSpec, cat=cat,
         batched_nms=batched_nms)

    _mod("detectron2.layers.nms", batched_nms_rotated=batched_nms_rotated)
    _mod("detectron2.structures", Boxes=Boxes, Instances=Instances, RotatedBoxes=RotatedBoxes)
    _mod("detectron2.utils")
    _mod("detectron2.utils.comm", get_world_size=lambda: 1, get_rank=lambda: 0, is_main_process=lambda: True, synchronize=lambda: None)
    _mod("detectron2.utils.env", TORCH_VERSION=tuple(int(x) for x in torch.__version__.split(".")[:2]))
    _mod("detectron2.modeling")
    _mod("detectron2.modeling.meta_arch")
    meta = Registry("META_ARCH")
    bb = Registry("BACKBONE")
    _mod("detectron2.modeling.meta_arch.build", META_ARCH_REGISTRY=meta)
    _mod("detectron2.modeling.backbone", BACKBONE_REGISTRY=bb, FPN=FPN, Backbone=Backbone)
    _mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=bb)
    _mod("detectron2.modeling.backbone.fpn", FPN=FPN, LastLevelMaxPool=LastLevelMaxPool, LastLevelP6P7=LastLevelP6P7)
    _mod("detectron2.modeling.postprocessing", detector_postprocess=detector_postprocess)

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    _mod("fvcore")
    _mod("fvcore.nn", sigmoid_focal_loss=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("training only")))
    _mod("fvcore.nn.weight_init", c2_msra_fill=c2_msra_fill, c2_xavier_fill=_xavier)
    _mod("pytorch3d")
    _mod("pytorch3d.transforms")
    _mod("pytorch3d.transforms.rotation_conversions", quaternion_to_matrix=quaternion_to_matrix, matrix_to_quaternion=matrix_to_quaternion)
    t3d = _mod("pytorch3d.transforms.transform3d", Translate=Translate, Rotate=Rotate, Transform3d=Transform3d)
    sys.modules["pytorch3d.transforms"].transform3d = t3d
    _mod("fvcore.nn.smooth_l1_loss", smooth_l1_loss=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("training only")))
    _mod("pyquaternion", Quaternion=Quaternion)
    _mod("mpi4py", MPI=types.SimpleNamespace(COMM_WORLD=None))
    _mod("cv2", INTER_NEAREST=0, INTER_LINEAR=1, INTER_CUBIC=2)
    _mod("fvcore.transforms", NoOpTransform=NoOpTransform)
    _mod("fvcore.transforms.transform", HFlipTransform=HFlipTransform, VFlipTransform=VFlipTransform, NoOpTransform=NoOpTransform,
         Transform=Transform, TransformList=TransformList)
    _mod("detectron2.data")
    _mod("detectron2.data.detection_utils", read_image=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("read_image")))
    _mod("detectron2.data.transforms", RandomFlip=D2RandomFlip, ResizeShortestEdge=D2ResizeShortestEdge, ResizeTransform=ResizeTransform,
         apply_augmentations=apply_augmentations)
    # tridet packages as bare namespaces (their __init__ chains pull in data / TTA / visualisation code)
    for pkg in ("tridet", "tridet.modeling", "tridet.modeling.dd3d", "tridet.utils", "tridet.structures"):
        m = _mod(pkg)
        m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
    for pkg in ("tridet.data", "tridet.data.datasets", "tridet.data.datasets.nuscenes"):
        _mod(pkg)
    _mod("tridet.data.datasets.nuscenes.build", MAX_NUM_ATTRIBUTES=3)  # tridet/data/datasets/nuscenes/build.py:77 (needs the devkit)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return d2
