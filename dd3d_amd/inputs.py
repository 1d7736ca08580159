"""Input side of the forward path on the device: what the reference's DefaultDatasetMapper does to a test
image before ``model(batched_inputs)`` (tridet/data/dataset_mappers/dataset_mapper.py:100-201, inference branch):
ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST) on the uint8 image + the matching intrinsics scaling
(tridet/data/augmentations/resize_transform.py:13-21,85-88).  The image resize is Pillow's 8-bit bilinear resampling ([ext]: detectron2
ResizeTransform.apply_image -> PIL Image.resize), reproduced bit-exactly by ``dd3d_resize_bilinear_u8``; the coefficient tables are
computed here in double precision, as Pillow does on the host.
"""
import ctypes as C
import math

import numpy as np
import torch

from dd3d_amd import hip

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """Bounds and fixed-point coefficients of Pillow's bilinear (triangle) resampling of `in_size` samples to `out_size`:
    (lo int32 [out], cnt int32 [out], kk int32 [out][ksize]).  The filter support grows with the down-scale factor (antialiasing)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale
    ksize = int(math.ceil(support)) * 2 + 1
    lo = np.zeros(out_size, dtype=np.int32)
    cnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        first = max(int(center - support + 0.5), 0)
        last = min(int(center + support + 0.5), in_size)
        w = []
        tot = 0.0
        for x in range(last - first):
            a = abs((x + first - center + 0.5) * inv)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            tot += v
        for x, v in enumerate(w):
            if tot != 0.0:
                v = v / tot
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        lo[xx], cnt[xx] = first, last - first
    return lo, cnt, kk


def shortest_edge_size(h, w, short_edge, max_size):
    """[ext] detectron2 ResizeShortestEdge.get_output_shape."""
    scale = short_edge * 1.0 / min(h, w)
    newh, neww = (short_edge, scale * w) if h < w else (scale * h, short_edge)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_intrinsics(K, h, w, new_h, new_w):
    """apply_imresize_intrinsics (resize_transform.py:13-21)."""
    K = torch.as_tensor(K, dtype=torch.float32)
    assert K.shape == (3, 3) and float(K[0, 1]) == 0 and torch.allclose(K, torch.triu(K))
    return K * torch.tensor([[new_w / w], [new_h / h], [1.0]], dtype=torch.float32)


class DeviceResizer:
    """Resizes uint8 (C, H, W) device images with the coefficient tables cached per (in, out) size."""
    def __init__(self, device):
        self.device = torch.device(device)
        self._tables = {}

    def _table(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            lo, cnt, kk = resample_coeffs(n_in, n_out)
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device) for a in (lo, cnt, kk)) + (kk.shape[1], )
        return self._tables[key]

    def __call__(self, img, new_h, new_w, out=None):
        """img: uint8 (C, H, W) on the device (any strides over H, W planes); out: optional uint8 (C, >=new_h, >=new_w) view to fill
        (e.g. a slot of ForwardPlan.in_u8).  Returns the (C, new_h, new_w) result view."""
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.is_cuda and img.stride(2) == 1
        Cc, H, W = img.shape
        if out is None:
            out = torch.empty((Cc, new_h, new_w), dtype=torch.uint8, device=img.device)
        assert out.dtype == torch.uint8 and out.stride(2) == 1 and out.shape[0] == Cc and out.shape[1] >= new_h and out.shape[2] >= new_w
        a = hip.ResizeArgs()
        a.src, a.dst = img.data_ptr(), out.data_ptr()
        a.C, a.H, a.W, a.new_h, a.new_w = Cc, H, W, new_h, new_w
        a.src_plane, a.src_row, a.dst_plane, a.dst_row = img.stride(0), img.stride(1), out.stride(0), out.stride(1)
        keep = []
        if new_w != W:
            lo, cnt, kk, ks = self._table(W, new_w)
            a.lo_w, a.cnt_w, a.kk_w, a.ksize_w = lo.data_ptr(), cnt.data_ptr(), kk.data_ptr(), ks
        if new_h != H:
            lo, cnt, kk, ks = self._table(H, new_h)
            a.lo_h, a.cnt_h, a.kk_h, a.ksize_h = lo.data_ptr(), cnt.data_ptr(), kk.data_ptr(), ks
        if new_w != W and new_h != H:
            tmp = torch.empty((Cc, H, new_w), dtype=torch.uint8, device=img.device)
            keep.append(tmp)
            a.tmp = tmp.data_ptr()
        with torch.cuda.device(img.device):
            hip.check(hip.lib().dd3d_resize_bilinear_u8(C.byref(a), hip.current_stream()), "resize_bilinear_u8")
        if keep:
            keep[0].record_stream(torch.cuda.current_stream(img.device))
        return out[:, :new_h, :new_w]


class DeviceInputMapper:
    """Test-time DefaultDatasetMapper on the device: raw uint8 BGR (3, H, W) image + raw intrinsics -> the input dict of
    ``model(batched_inputs)`` ("image" resized on the device, "intrinsics" scaled, "height"/"width" = the original size)."""
    def __init__(self, cfg, device="cuda"):
        self.min_size = int(cfg.INPUT.RESIZE.MIN_SIZE_TEST)
        self.max_size = int(cfg.INPUT.RESIZE.MAX_SIZE_TEST)
        self.resize = DeviceResizer(device)

    def __call__(self, image, intrinsics=None, **extra):
        image = image.to(self.resize.device, non_blocking=True)
        _, h, w = image.shape
        new_h, new_w = shortest_edge_size(h, w, self.min_size, self.max_size)
        d = dict(extra)
        d["image"] = self.resize(image, new_h, new_w)
        d["height"], d["width"] = h, w
        if intrinsics is not None:
            d["intrinsics"] = resize_intrinsics(intrinsics, h, w, new_h, new_w)
        return d
