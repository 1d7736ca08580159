"""Activation buffers (f32 NHWC and / or split planes) and weight packing of the launch plans (dd3d_amd.engine)."""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm


# --------------------------------------------------------------------------------------------- buffers
class Buf:
    """Activation tensor [B, H, W, pitch channels] in HBM, held in one or both of two storages:
      f32     `.t`  NHWC fp32 [B, H, W, pitch]                        -- what the pooling / top-down / gating / decode kernels and the
                                                                         residual adds read
      planes  `.p`  int16 [pitch/32][B*H*W][NP][32]                   -- the split-plane form one convolution hands to the next
                                                                         (include/dd3d_hip.h); NP = 16-bit terms of the math mode
    A dry-run (CPU) plan always carries `.t`: it is the plan emulator's logical tensor, whatever the device storages would be."""
    def __init__(self, B, H, W, C, device, name="", f32=True, planes=0, dry_run=False, f16=False, plane_scale=1.0):
        self.B, self.H, self.W, self.pitch, self.name = B, H, W, C, name
        self.f16, self.plane_scale = bool(f16), float(plane_scale)  # terms are IEEE halves of value * plane_scale (DD3D_MATH_F16X2), else bf16
        self.has_f32, self.np = bool(f32) or not planes, int(planes)
        self.t = torch.zeros((B, H, W, C), dtype=torch.float32, device=device) if (self.has_f32 or dry_run) else None
        self.p = None
        if planes:
            assert C % 32 == 0, (name, C)
            if not dry_run:
                self.p = torch.zeros((C // 32, B * H * W, planes, 32), dtype=torch.int16, device=device)

    def view(self, c0=0, C=None):
        return View(self, c0, self.pitch - c0 if C is None else C)

    def nchw(self, c0=0, C=None):
        C = self.pitch - c0 if C is None else C
        if self.t is not None:
            return self.t[..., c0:c0 + C].permute(0, 3, 1, 2)
        # planes only: the value the planes encode (exact for the three-term split), channels c0 .. c0 + C
        k0, k1 = c0 // 32, (c0 + C + 31) // 32
        if self.f16:
            terms = self.p[k0:k1].view(torch.float16).float() / self.plane_scale
        else:
            terms = (self.p[k0:k1].to(torch.int32) << 16).view(torch.float32)  # [chunks][BHW][NP][32]
        x = terms[:, :, 0]
        for q in range(1, self.np):
            x = x + terms[:, :, q]
        x = x.permute(1, 0, 2).reshape(self.B, self.H, self.W, (k1 - k0) * 32)
        return x[..., c0 - 32 * k0:c0 - 32 * k0 + C].permute(0, 3, 1, 2)


class View:
    """Channel slice [c0, c0+C) of a Buf."""
    def __init__(self, buf, c0, C):
        assert c0 % 4 == 0 and 0 <= c0 and c0 + C <= buf.pitch, (c0, C, buf.pitch)
        assert not buf.np or c0 % 32 == 0, (buf.name, c0)  # a slice of a split-plane buffer is a run of whole 32-channel chunk images
        self.buf, self.c0, self.C = buf, c0, C

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + 4 * self.c0 if self.buf.has_f32 and self.buf.t is not None else 0

    @property
    def pptr(self):
        """First chunk image of the slice in the split-plane storage (0 when the buffer has none)."""
        b = self.buf
        return b.p.data_ptr() + (self.c0 // 32) * (b.B * b.H * b.W) * b.np * 64 if b.p is not None else 0

    B = property(lambda s: s.buf.B)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)
    pitch = property(lambda s: s.buf.pitch)
    has_f32 = property(lambda s: s.buf.has_f32)
    np = property(lambda s: s.buf.np)

    def nchw(self):
        return self.buf.nchw(self.c0, self.C)


def dense_filter(conv):
    """OIHW filter of a convolution as the dense kernels see it: a grouped convolution (BottleneckX, dla.py:118-128) becomes a
    block-diagonal filter, output group g reading input group g only.  Costs `groups` times the grouped FLOPs; no reference config uses
    a grouped layer, so no grouped kernel is built."""
    w = conv.weight.detach()
    g = getattr(conv, "groups", 1)
    if g == 1:
        return w
    O, Ig, KH, KW = w.shape
    dense = torch.zeros((O, Ig * g, KH, KW), dtype=w.dtype, device=w.device)
    og = O // g
    for k in range(g):
        dense[k * og:(k + 1) * og, k * Ig:(k + 1) * Ig] = w[k * og:(k + 1) * og]
    return dense


def pad32(c):
    return (c + 31) // 32 * 32


def scatter_in_channels(weight, segments):
    """OIHW filter whose input channels are the concatenation of `segments` = [(real, padded), ...] slices -> the filter for the buffer
    in which every slice is padded to `padded` channels (zero weights on the padding).  The implicit-GEMM kernels walk 32-channel
    chunks, so a concat buffer holding 80- or 112-channel slices (V-19-slim-eSE) keeps each slice 32-aligned and zero-padded."""
    O, I, KH, KW = weight.shape
    assert I == sum(r for r, _ in segments), (I, segments)
    out = torch.zeros((O, sum(p for _, p in segments), KH, KW), dtype=weight.dtype, device=weight.device)
    src = dst = 0
    for r, p in segments:
        out[:, dst:dst + r] = weight[:, src:src + r]
        src, dst = src + r, dst + p
    return out


# --------------------------------------------------------------------------------------------- weight packing
def pack_filter(weights, device):
    """OIHW filters (list => concatenated along O) -> Wp[Npad][Kpad], k = (c/CC)*(T*CC) + tap*CC + c%CC
    (include/dd3d_hip.h).  Returns (tensor, meta)."""
    w = torch.cat([x.detach().float().cpu() for x in weights], 0) if isinstance(weights, (list, tuple)) else weights.detach().float().cpu()
    N, Cin, KH, KW = w.shape
    cin_p = Cin
    if Cin < 32 and Cin not in (4, 16):
        cin_p = 4 if Cin <= 4 else 16 if Cin <= 16 else 32
    elif Cin > 32 and Cin % 32:
        cin_p = (Cin + 31) // 32 * 32
    if cin_p != Cin:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_p - Cin))
    CC = min(cin_p, 32)
    T = KH * KW
    wp = w.permute(0, 2, 3, 1).reshape(N, T, cin_p // CC, CC).permute(0, 2, 1, 3).reshape(N, T * cin_p)
    K = T * cin_p
    Kpad = (K + 31) // 32 * 32
    Npad = (N + 31) // 32 * 32
    out = torch.zeros((Npad, Kpad), dtype=torch.float32)
    out[:N, :K] = wp
    meta = dict(N=N, Cin=cin_p, KH=KH, KW=KW, Kpad=Kpad, Npad=Npad)
    return out.to(device), meta


def split_bf16x3(wp):
    """Wp[Npad][Kpad] f32 -> Wp3[Npad][Kpad/32][3][32] bf16 (int16 bit patterns): x = hi + mid + lo exactly, each term the
    next 8 significand bits (truncation), as conv_igemm_bf16x3_kernel splits the activations (include/dd3d_hip.h)."""
    x = wp.detach().float().cpu().contiguous()
    mask = torch.tensor(-65536, dtype=torch.int32)  # 0xffff0000
    hi = (x.view(torch.int32) & mask).view(torch.float32)
    r = x - hi
    mid = (r.view(torch.int32) & mask).view(torch.float32)
    lo = r - mid
    planes = torch.stack([hi, mid, lo], 0).view(torch.int32) >> 16  # arithmetic shift; the low 16 bits are what we keep
    planes = planes.to(torch.int16)  # wraps to the same 16-bit pattern
    Npad, Kpad = x.shape
    return planes.view(3, Npad, Kpad // 32, 32).permute(1, 2, 0, 3).contiguous()


def split_planes_host(wp, math):
    """Wp[Npad][Kpad] f32 -> [Npad][Kpad/32][NP][32] 16-bit terms of arithmetic mode `math`, split exactly as the kernels split the
    activations (csrc/conv_common.h::split_pack): three truncated bf16 terms (BF16X3), or round-to-nearest-even hi (+ lo) terms."""
    if math == hip.MATH_BF16X3:
        return split_bf16x3(wp)
    x = wp.detach().float().cpu().contiguous()
    if math == hip.MATH_F16X2:
        raise ValueError("the half-term split carries a per-row scale: use split_f16x2_host")
    hi = x.to(torch.bfloat16)
    terms = [hi]
    if math == hip.MATH_BF16X2:
        terms.append((x - hi.float()).to(torch.bfloat16))
    Npad, Kpad = x.shape
    planes = torch.stack([t.view(torch.int16) for t in terms], 0)
    return planes.view(len(terms), Npad, Kpad // 32, 32).permute(1, 2, 0, 3).contiguous()


def split_f16x2_host(wp):
    """Wp[Npad][Kpad] f32 -> ([Npad][Kpad/32][2][32] IEEE-half terms of Wp[n] * s[n], s[Npad]): hi = half(x s), lo = half(x s - hi), both
    round-to-nearest (csrc/conv_common.h::split_pack).  s[n] is the power of two that brings the largest |Wp[n, :]| into [2^13, 2^14):
    hi then carries 11 bits and lo the next 11 wherever |x s| >= 2^-2, and the absolute floor 2^-25 / s[n] sits ~2^-39 below the row's
    largest filter tap.  The caller divides s[n] (exactly) out of the epilogue scale."""
    x = wp.detach().float().cpu().contiguous()
    amax = x.abs().amax(1)
    e = torch.floor(torch.log2(amax.clamp(min=1e-30)))
    s = torch.where(amax > 0, torch.exp2(13.0 - e), torch.ones_like(amax))
    y = x * s[:, None]
    hi = y.to(torch.float16)
    lo = (y - hi.float()).to(torch.float16)
    Npad, Kpad = x.shape
    planes = torch.stack([hi.view(torch.int16), lo.view(torch.int16)], 0)
    return planes.view(2, Npad, Kpad // 32, 32).permute(1, 2, 0, 3).contiguous(), s


def pack_smallc_bf16x3(weights, cin_p):
    """OIHW filter (Cin <= cin_p in {4, 16}) -> [chunk][plane][Npad16][32] bf16 bit patterns in the k order of
    dd3d_conv2d_smallc_bf16x3 (include/dd3d_hip.h)."""
    w = weights.detach().float().cpu()
    N, Cin, KH, KW = w.shape
    n16 = (N + 15) // 16 * 16
    if cin_p == 4:
        k = torch.zeros((n16, KH, 8, 4))
        k[:N, :, :KW, :Cin] = w.permute(0, 2, 3, 1)
        k = k.reshape(n16, KH, 32)  # chunk = filter row
    else:
        T = KH * KW
        k = torch.zeros((n16, (T + 1) // 2 * 2, 16))
        k[:N, :T, :Cin] = w.permute(0, 2, 3, 1).reshape(N, T, Cin)
        k = k.reshape(n16, (T + 1) // 2, 32)  # chunk = two taps
    planes = split_bf16x3(k.reshape(n16, -1))  # [n16][chunks][3][32]
    return planes.permute(1, 2, 0, 3).contiguous()


def pack_smallc_f16x2(weights, cin_p):
    """OIHW filter (Cin <= cin_p in {4, 16}) -> ([chunk][plane hi, lo][Npad16][32] IEEE-half bit patterns, row scales s[Npad16]) in the k
    order of the patch kernels (pack_smallc_bf16x3 / include/dd3d_hip.h::dd3d_stem_args); the terms are those of w[n] * s[n], split as
    split_f16x2_host splits every other filter of the two-half-term arithmetic."""
    w = weights.detach().float().cpu()
    N, Cin, KH, KW = w.shape
    n16 = (N + 15) // 16 * 16
    if cin_p == 4:
        k = torch.zeros((n16, KH, 8, 4))
        k[:N, :, :KW, :Cin] = w.permute(0, 2, 3, 1)
        k = k.reshape(n16, KH * 32)  # chunk = filter row
    else:
        T = KH * KW
        k = torch.zeros((n16, (T + 1) // 2 * 2, 16))
        k[:N, :T, :Cin] = w.permute(0, 2, 3, 1).reshape(N, T, Cin)
        k = k.reshape(n16, (T + 1) // 2 * 32)  # chunk = two taps
    planes, s = split_f16x2_host(k)  # [n16][chunks][2][32]
    return planes.permute(1, 2, 0, 3).contiguous(), s

