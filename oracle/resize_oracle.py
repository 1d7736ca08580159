"""CPU ORACLE for the input-side image resize (test infrastructure only).

The reference resizes with detectron2's ResizeTransform.apply_image [ext] = PIL ``Image.resize(size, BILINEAR)`` on the uint8 HWC
image (tridet/data/augmentations/resize_transform.py:85-88 -> detectron2.data.transforms.ResizeTransform; size rule
ResizeShortestEdge [ext]).  Pillow is installed here, so the restatement below (Pillow's libImaging/Resample.c: separable
convolution, coefficients from a triangle filter whose support grows with the down-scale factor, 22-bit fixed point, an 8-bit
intermediate after the horizontal pass) is PINNED against the real library in tests/test_resize.py.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the full input range.
    Returns (xmin int32 [out], xsize int32 [out], kk int32 [out][ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    xsize = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        x = np.arange(n, dtype=np.float64)
        arg = np.abs((x + lo - center + 0.5) * ss)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        tot = 0.0
        for v in w:  # same summation order as the C loop
            tot += float(v)
        if tot != 0.0:
            w = w / tot
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS))).astype(np.int64)  # C (int) truncation
        xmin[xx], xsize[xx] = lo, n
        kk[xx, :n] = q.astype(np.int32)
    return xmin, xsize, kk


def _pass(img, xmin, xsize, kk, axis):
    """One pass along `axis` of a (C, H, W) uint8 array: clip8((2^21 + sum in * kk) >> 22)."""
    img = np.moveaxis(img.astype(np.int64), axis, -1)
    out = np.empty(img.shape[:-1] + (len(xmin), ), dtype=np.uint8)
    for xx in range(len(xmin)):
        lo, n = int(xmin[xx]), int(xsize[xx])
        acc = (1 << (PRECISION_BITS - 1)) + (img[..., lo:lo + n] * kk[xx, :n].astype(np.int64)).sum(-1)
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_bilinear_u8(img_chw, new_h, new_w):
    """(C, H, W) uint8 -> (C, new_h, new_w) uint8, horizontal pass first (ImagingResample), a pass is skipped when its size is kept."""
    C, H, W = img_chw.shape
    out = img_chw
    if new_w != W:
        out = _pass(out, *resample_coeffs(W, new_w), axis=2)
    if new_h != H:
        out = _pass(out, *resample_coeffs(H, new_h), axis=1)
    return out


def shortest_edge_size(h, w, short_edge, max_size):
    """[ext] detectron2 ResizeShortestEdge.get_output_shape (v0.5/0.6)."""
    scale = short_edge * 1.0 / min(h, w)
    if h < w:
        newh, neww = short_edge, scale * w
    else:
        newh, neww = scale * h, short_edge
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def resize_intrinsics(K, h, w, new_h, new_w):
    """apply_imresize_intrinsics, tridet/data/augmentations/resize_transform.py:13-21."""
    K = np.asarray(K, dtype=np.float32)
    assert K.shape == (3, 3) and K[0, 1] == 0 and np.allclose(K, np.triu(K))
    return K * np.float32([new_w / w, new_h / h, 1]).reshape(3, 1)
