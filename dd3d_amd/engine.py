"""Launch-plan builder and executor of the DD3D forward path on MI355X.

``ForwardPlan`` walks the parameter tree of a ``DD3D`` model once per input geometry (B, Hp, Wp), packs the
weights (filters re-ordered for the implicit-GEMM K order, norms folded into per-channel scale/shift), lays
out every activation as an NHWC fp32 buffer in HBM and records the sequence of libdd3d_hip launches.
``run()`` replays that sequence on the current HIP stream -- either launch by launch or as one captured
hipGraph -- with no host synchronisation between the uint8 image and the final detection buffer.

PyTorch is used here for device memory, streams and graph capture only; all arithmetic is in
dd3d_amd/csrc (C ABI: include/dd3d_hip.h).

Reference behaviour being reproduced: tridet/modeling/dd3d/core.py:64-164 (DD3D.forward, inference branch).
"""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm

NUM_CU = 256  # MI355X


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


class FusedStemOp:
    """One dd3d_stem_fused_f16x2 launch: uint8 image -> normalise -> base_layer -> level0 -> level1 (DLA, dla.py:271-280,327-344), the
    intermediate maps kept in LDS.  Replaces preprocess + three convolutions + the plane split of level1's output."""
    def __init__(self, plan, model, convs, vout, name="stem"):
        from dd3d_amd.layers import fold_norm
        assert plan.math == hip.MATH_F16X2
        self.name, self.macs = name, 0
        B, Hp, Wp = plan.B, plan.Hp, plan.Wp
        a = hip.StemArgs()
        self.keep = []
        descs = []
        for i, (conv, cin_p) in enumerate(zip(convs, (4, 16, 16)), 1):
            planes, row_scale = pack_smallc_f16x2(conv.weight, cin_p)
            scale, shift = fold_norm(conv, None)
            n = conv.out_channels
            # acc = (S x) . (s[n] w): both power-of-two scales leave through the epilogue scale, exactly
            sc = (scale.detach().float().cpu() / (row_scale[:n] * float(plan.act_scale))).to(plan.device)
            bi, wdev = plan._vec(shift), planes.to(plan.device)
            setattr(a, f"w{i}", wdev.data_ptr())
            setattr(a, f"scale{i}", sc.data_ptr())
            setattr(a, f"bias{i}", bi.data_ptr())
            self.keep += [wdev, sc, bi]
            descs.append(dict(weight=conv.weight, stride=conv.stride, pad=conv.padding, scale=plan._vec(scale), bias=bi))
            Ho, Wo = (Hp, Wp) if i < 3 else (Hp // 2, Wp // 2)
            self.macs += B * Ho * Wo * n * conv.weight.shape[1] * conv.weight.shape[2] * conv.weight.shape[3]
        a.src, a.sizes = plan.in_u8.data_ptr(), plan.in_sizes.data_ptr()
        for c in range(3):
            a.mean[c], a.stdv[c] = float(model.pixel_mean.flatten()[c]), float(model.pixel_std.flatten()[c])
        a.out = vout.ptr if vout.has_f32 else None
        a.out_planes = vout.pptr if vout.np else None
        a.B, a.Hp, a.Wp, a.out_pitch = B, Hp, Wp, vout.pitch
        a.plane_scale = float(plan.act_scale)
        a.status = plan.status.data_ptr()
        self.a = a
        self.desc = dict(kind="fused_stem", convs=descs, vout=vout, mean=[float(v) for v in model.pixel_mean.flatten()],
                         std=[float(v) for v in model.pixel_std.flatten()], planes=bool(vout.np))
        self.info = dict(name=name, M=B * (Hp // 2) * (Wp // 2), N=32, K=0, tile="fused", splitk=1, math=hip.MATH_F16X2, blocks=0, nsegs=1)

    def __call__(self, lib, stream):
        hip.check(lib.dd3d_stem_fused_f16x2(C.byref(self.a), stream), "fused stem " + self.name)


class SmallcConvOp:
    """One dd3d_conv2d_smallc_bf16x3 launch: a stem convolution fed from an LDS patch (no im2col loop)."""
    def __init__(self, plan, conv_weight, cin_p, stride, pad, vin, vout, scale, bias, relu, name=""):
        N, _, KH, KW = conv_weight.shape
        self.name = name
        self.w3 = pack_smallc_bf16x3(conv_weight, cin_p).to(plan.device)
        self.keep = [scale, bias]
        self.desc = dict(kind="smallc_conv", weight=conv_weight, stride=stride, pad=pad, vin=vin, vout=vout, scale=scale, bias=bias, relu=bool(relu))
        a = hip.SmallcArgs()
        a.in_, a.out, a.w3 = vin.ptr, vout.ptr, self.w3.data_ptr()
        a.scale, a.bias, a.lo = scale.data_ptr(), bias.data_ptr(), None
        a.B, a.H, a.W, a.Ho, a.Wo = vin.B, vin.H, vin.W, vout.H, vout.W
        a.in_pitch, a.out_pitch = vin.pitch, vout.pitch
        a.Cin, a.KH, a.KW, a.stride, a.pad, a.N, a.relu = cin_p, KH, KW, stride, pad, N, int(relu)
        self.a = a
        M = vout.B * vout.H * vout.W
        self.macs = M * N * KH * KW * cin_p
        self.info = dict(name=name, M=M, N=N, K=KH * KW * cin_p, tile="patch", splitk=1, math=hip.MATH_BF16X3, blocks=0, nsegs=1)

    def __call__(self, lib, stream):
        hip.check(lib.dd3d_conv2d_smallc_bf16x3(C.byref(self.a), stream), "smallc conv " + self.name)


MATH_TILES = {  # tile configurations instantiated per arithmetic mode
    hip.MATH_F32: (hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x64, hip.TILE_128x32, hip.TILE_64x128),
    hip.MATH_BF16X3: (hip.TILE_256x128, hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x128, hip.TILE_128x128_W4, hip.TILE_64x64_W4,
                      hip.TILE_128x64_W4, hip.TILE_128x64_K2, hip.TILE_64x128_K2, hip.TILE_64x64_W4K2),
}
# the split-plane kernel (csrc/conv_planes.hip): one barrier per K-tile for every tile, so no "two K-tiles per barrier" variants
PLANE_TILES = (hip.TILE_256x128, hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x128, hip.TILE_128x128_W4, hip.TILE_64x64_W4, hip.TILE_128x64_W4,
               hip.TILE_256x128_T42, hip.TILE_128x256_T24, hip.TILE_256x256_W8, hip.TILE_128x32_W4)
BIG_WAVE_TILES = (hip.TILE_256x128_T42, hip.TILE_128x256_T24, hip.TILE_256x256_W8)  # 8 accumulator blocks per wave; picked by the measured table only
PLANE_TILE_ALIAS = {hip.TILE_128x64_K2: hip.TILE_128x64, hip.TILE_64x128_K2: hip.TILE_64x128, hip.TILE_64x64_W4K2: hip.TILE_64x64_W4}
for _m in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2):
    MATH_TILES[_m] = PLANE_TILES
# blocks of a configuration that can share a CU (LDS-limited); the f32 kernels were measured, see profiles/
BLOCKS_PER_CU = {hip.TILE_128x128_W4: 2, hip.TILE_64x64_W4: 2, hip.TILE_128x64_W4: 2}  # tiles the split-bf16 kernel is instantiated for


MATH_NAMES = {"f32": hip.MATH_F32, "bf16x3": hip.MATH_BF16X3, "bf16x2": hip.MATH_BF16X2, "bf16": hip.MATH_BF16, "f16x2": hip.MATH_F16X2}


def default_math():
    """Arithmetic of the Cin % 32 == 0 convolutions (all accumulate in f32).  Env DD3D_MATH or model.math.
      f32-equivalent (measured against a float64 convolution they sit at the same ~3e-7 as exact f32, tests/test_conv_planes_gpu.py):
        "f16x2"  (default) two IEEE-half terms per operand, 3 cross products on the f16 matrix pipe; needs |activation| <= 65504 /
                 plane scale -- a kernel-side status word trips otherwise and the forward raises / falls back to "bf16x3"
        "bf16x3" three bf16 terms, 6 cross products; the full f32 exponent range
        "f32"    v_mfma_f32_32x32x2_f32, bitwise an fmaf chain (1/16 of the bf16 rate)
      reduced (what BASELINE.json's bf16 configurations name):
        "bf16x2" two bf16 terms, 3 products (~1e-5 relative);  "bf16" plain bf16 operands (~1e-2: misses the 1e-3 parity bar)"""
    return MATH_NAMES[os.environ.get("DD3D_MATH", "f16x2")]


# (TM, TN, WM, WN) of the split-plane kernels' tiles: 32 x 32 accumulator blocks per wave and the block's wave grid
TILE_WAVE_GRID = {hip.TILE_256x128: (2, 2, 4, 2), hip.TILE_128x128: (2, 1, 2, 4), hip.TILE_128x64: (1, 1, 4, 2), hip.TILE_64x128: (1, 1, 2, 4),
                  hip.TILE_128x128_W4: (2, 2, 2, 2), hip.TILE_64x64_W4: (1, 1, 2, 2), hip.TILE_128x64_W4: (2, 1, 2, 2),
                  hip.TILE_256x128_T42: (4, 2, 2, 2), hip.TILE_128x256_T24: (2, 4, 2, 2), hip.TILE_256x256_W8: (4, 2, 2, 4),
                  hip.TILE_128x32_W4: (1, 1, 4, 1)}


def kernel_signature(op):
    """Name of the kernel instantiation a ConvOp launches, as rocprofv3 prints it (bench.py / profiles bookkeeping)."""
    cfg = op.L.tile_cfg
    tm_tn_wm_wn = TILE_WAVE_GRID
    sk = "true" if op.L.splitk > 1 else "false"
    if op.in_planes:
        tm, tn, wm, wn = tm_tn_wm_wn[PLANE_TILE_ALIAS.get(cfg, cfg)]
        np_ = hip.MATH_PLANES[op.math]
        bm, bn = tm * 32 * wm, tn * 32 * wn
        L = op.L
        nk = L.Kpad // 32
        row = (L.KH == 3 and L.KW == 3 and L.stride == 1 and L.pad == 1 and (L.splitk == 1 or -(-nk // L.splitk) % 3 == 0)
               and os.environ.get("DD3D_CONV_ROW", "1") != "0")
        if row:  # csrc/conv_planes_row.hip: the three taps of a filter row share one A stage
            # ring depths: what the LIBRARY instantiates (dd3d_conv_row_rings: they are build-time properties of the .so); the formula below
            # (csrc/conv_planes_row.hip::RowRings with the product build's defaults) only serves a box without the library
            nsb = nsa = None
            try:
                b_, a_ = C.c_int32(), C.c_int32()
                if hip.lib().dd3d_conv_row_rings(PLANE_TILE_ALIAS.get(cfg, cfg), op.math, C.byref(b_), C.byref(a_)) == 0:
                    nsb, nsa = b_.value, a_.value
            except (hip.HipLibraryMissing, OSError):
                pass
            if nsb is None:
                nsb, nsa = row_rings_default(np_, bm, bn, wm * wn)
            return f"dd3d::conv_igemm_planes_row_kernel<{tm}, {tn}, {wm}, {wn}, {nsb}, {op.math}, {sk}, {nsa}>"
        stage = np_ * (bm + bn) * 64
        ns = max(2, min(4, ((144 if (wm * wn == 8 or stage > 32768) else 72) * 1024) // stage))
        return f"dd3d::conv_igemm_planes_kernel<{tm}, {tn}, {wm}, {wn}, {ns}, {op.math}, {sk}, 0>"
    if op.math == hip.MATH_BF16X3:
        return f"dd3d::conv_igemm_bf16x3_kernel ({hip.TILE_NAMES[cfg]}, split-K {sk})"
    return f"dd3d::conv_igemm_f32[_dma]_kernel ({hip.TILE_NAMES[cfg]}, split-K {sk})"


def row_rings_default(np_, bm, bn, nwaves):
    """(NSB, NSA) of csrc/conv_planes_row.hip::RowRings for the product build (no -DDD3D_ROW_* knob); a CPU test compares it with
    dd3d_conv_row_rings for every tile and mode."""
    ast, bst = np_ * (bm + 16) * 64, np_ * bn * 64
    budget = (152 if (nwaves == 8 or 2 * ast > 65536) else 76) * 1024
    nsb = 3 if 2 * ast + 3 * bst <= budget else 2
    return nsb, 2


def tile_key(m_list, N, Kpad, stride):
    return f"{'+'.join(str(m) for m in m_list)},{N},{Kpad},{stride}"


def _load_tile_table(math_name):
    """Measured exceptions to the analytic model below: {tile_key: [tile, splitk, best_us, model_us]}, produced on an MI355X by
    tests/gpu_tile_explore.py (every candidate timed; entries kept only where the best beats the model's pick by > 5 %)."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "data", f"tile_table_{math_name}.json")
    return json.load(open(path)) if os.path.exists(path) else {}


TILE_TABLE = {hip.MATH_F32: _load_tile_table("f32"), hip.MATH_BF16X3: _load_tile_table("bf16x3"), hip.MATH_BF16X2: _load_tile_table("bf16x2"),
              hip.MATH_BF16: _load_tile_table("bf16"), hip.MATH_F16X2: {}}
PLANE_TILE_TABLE = {m: _load_tile_table(n + "_planes") for n, m in (("bf16x3", hip.MATH_BF16X3), ("bf16x2", hip.MATH_BF16X2), ("bf16", hip.MATH_BF16),
                                                                     ("f16x2", hip.MATH_F16X2))}


def _tile_overrides():
    """DD3D_TILE_OVERRIDE: `key=tile:splitk` pairs separated by ';' (key as `tile_key` prints it) that win over the measured table --
    for sweeps of the issue mode, where the tile that minimises one launch's latency need not maximise the throughput of several slots."""
    spec = os.environ.get("DD3D_TILE_OVERRIDE", "")
    out = {}
    for item in filter(None, (s.strip() for s in spec.split(";"))):
        key, val = item.split("=")
        tile, _, sk = val.partition(":")
        out[key.strip()] = [tile.strip(), int(sk or 1)]
    return out


def choose_tiling(m_list, N, Kpad, stride=1, math=0, planes=False):
    """Pick (tile_cfg, splitk): a measured table entry when this exact shape has one, else minimise the modelled makespan
    on 256 CUs: every block costs BM*BN*K MACs on its CU's matrix pipe (partial tiles cost the same as full ones); split-K
    adds the partial-sum exchange.  `planes`: the split-plane-input kernel (its own measured table; the f32-input kernel's
    entries serve as the fallback for the three-term mode, their two-K-tiles-per-barrier variants mapped to the plain tile)."""
    allowed = PLANE_TILES if planes else MATH_TILES[math]
    # (measured and dropped: forcing the small convolutions onto 4-wave blocks with <= 48 KiB of LDS so that other streams' blocks could
    # share their CUs -- pipelined throughput 960 -> 921 img/s, profiles/r02_notes.md)
    key = tile_key(m_list, N, Kpad, stride)
    hit = PLANE_TILE_TABLE[math].get(key) if planes else TILE_TABLE[math].get(key)
    forced = _tile_overrides().get(key)  # DD3D_TILE_OVERRIDE="M[+M..],N,Kpad,stride=tile:splitk;..." (measurement sweeps, tests/gpu_issue_sweep.sh)
    if forced is not None:
        hit = forced
    if hit is None and planes and math == hip.MATH_BF16X3:
        hit = TILE_TABLE[math].get(key)
    if hit is not None:
        cfg = next(c for c, nm in hip.TILE_NAMES.items() if nm == hit[0])
        return (PLANE_TILE_ALIAS.get(cfg, cfg) if planes else cfg), int(hit[1])
    nk = Kpad // 32
    # Large launches of the two-term modes with N >= 256: the 8-wave 256 x 256 tile (wave tile 128 x 64: one ds_read_b128 per two MFMAs) beat
    # every other tile by 7-9 % on every such shape measured in round 4 -- head towers, the merged FPN output launch, V2-99's 1 x 1 concat
    # convolutions, from 158 blocks (profiles/r04h_*, r04t_*) -- so it is the default there, not only where a table entry names it.
    if planes and hip.MATH_PLANES[math] <= 2 and N >= 256 and hip.TILE_256x256_W8 in allowed and -(-N // 256) * 256 <= 1.15 * N:
        if sum(-(-m // 256) for m in m_list) * -(-N // 256) >= 150:
            return hip.TILE_256x256_W8, 1
    best = None
    for cfg in allowed:
        if cfg in BIG_WAVE_TILES or cfg == hip.TILE_128x32_W4:
            continue  # (no analytic model: the measured table or an explicit `tile=` selects them)
        bm, bn = hip.TILE_SHAPES[cfg]
        if bn == 32 and N > 32:
            continue
        if bn > 32 and N <= 32 and not (planes and bn == 64):  # (the split-plane kernel has no 32-wide tile: narrow convs pad to 64)
            continue
        if bn == 128 and N <= 64:
            continue
        blocks = sum(-(-m // bm) for m in m_list) * -(-N // bn)
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (nk // sk < 4 or blocks >= NUM_CU):
                continue
            per = -(-nk // sk)
            cost = -(-blocks * sk // NUM_CU) * bm * bn * per * 32
            # measured matrix-pipe efficiency of each tile shape once the chip is full (profiles/r01b_conv_ops.txt):
            # 128x128 ~119 TF/s, 128x64 ~95, 64x64 ~74, 128x32 (N <= 32 pads the 32-wide MFMA) ~45
            if math == hip.MATH_F32:
                cost /= {(128, 128): 1.0, (128, 64): 0.80, (64, 128): 0.80, (64, 64): 0.63, (128, 32): 0.40}[(bm, bn)]
            else:  # split-bf16 kernel: ~2x the f32 rate on the big tiles, LDS-read bound on the small ones
                cost /= {hip.TILE_256x128: 2.4, hip.TILE_128x128: 1.8, hip.TILE_128x64: 1.3, hip.TILE_64x128: 1.3,
                         hip.TILE_128x128_W4: 1.0, hip.TILE_64x64_W4: 0.6, hip.TILE_128x64_W4: 0.8,
                         hip.TILE_128x64_K2: 1.0, hip.TILE_64x128_K2: 1.0, hip.TILE_64x64_W4K2: 0.5}[cfg]  # rough; the table decides
                if math in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2):  # fewer products per K-tile: the matrix term shrinks, the rest does not
                    cost *= {hip.MATH_BF16X2: 0.6, hip.MATH_F16X2: 0.6, hip.MATH_BF16: 0.35}[math]
            if sk > 1:
                # second launch (~2 us) + partial-sum round trip (sk*M*N*8 B at ~3 TB/s), in per-CU MAC units
                # (one CU retires 157.3e12 / 2 / 256 = 3.07e11 MAC/s)
                cost += 0.6e6 + sk * sum(m_list) * N * 8 / 3e12 * 3.07e11
            if best is None or cost < best[0]:
                best = (cost, cfg, sk)
    return best[1], best[2]


class ConvOp:
    """One dd3d_conv2d_igemm_f32 launch (possibly many segments).  Input form: the split planes of the input buffers when they have
    them (plan.use_planes), else f32 NHWC.  Output form per segment: every storage its output buffer has (f32 NHWC and / or split
    planes), unless the segment says `write_f32=False` / `write_planes=False`."""
    def __init__(self, plan, meta, stride, pad, segs, relu, tile=None, splitk=None, name="", math=None, in_relu=False):
        dev = plan.device
        self.name = name
        m_list = [s["out"].B * s["out"].H * s["out"].W for s in segs]
        if math is None:
            math = plan.math
        in_planes = math != hip.MATH_F32 and meta["Cin"] % 32 == 0 and all(s["in"].np == hip.MATH_PLANES[math] for s in segs) and not in_relu
        if meta["Cin"] % 32 or (meta["N"] <= 32 and not in_planes):  # stem layers (Cin 4 / 16) and narrow convs on f32 input: the f32 kernel
            math = hip.MATH_F32
        if math in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2) and not in_planes:
            raise ValueError(f"conv {name}: math mode {math} reads split-plane input only; its input buffer has none")
        self.math, self.in_planes = math, in_planes
        cfg, sk = choose_tiling(m_list, meta["N"], meta["Kpad"], stride, math, planes=in_planes)
        if tile is not None:
            if tile not in (PLANE_TILES if in_planes else MATH_TILES[math]):
                raise ValueError(f"conv {name}: tile {hip.TILE_NAMES[tile]} is not instantiated for math mode {math}")
            cfg = tile
        if splitk is not None:
            sk = splitk
        if cfg == hip.TILE_256x256_W8 and any(sg.get("res") is not None for sg in segs):
            cfg = hip.TILE_256x128  # (the 8-wave 256 x 256 tile has no registers left for a residual in flight)
        bm, bn = hip.TILE_SHAPES[cfg]
        arr = np.zeros(len(segs), dtype=hip.CONV_SEG_DTYPE)
        tiles = []
        self.keep = []
        self.out_forms = []
        self.res_forms = []
        for i, s in enumerate(segs):
            vin, vout = s["in"], s["out"]
            assert vin.C == meta["Cin"], (name, vin.C, meta["Cin"])
            assert vout.C >= (s.get("n_limit") or meta["N"]), (name, vout.C, meta["N"])
            Ho = (vin.H + 2 * pad - meta["KH"]) // stride + 1
            Wo = (vin.W + 2 * pad - meta["KW"]) // stride + 1
            assert (Ho, Wo) == (vout.H, vout.W) and vin.B == vout.B, (name, Ho, Wo, vout.H, vout.W)
            a = arr[i]
            w = s["w"] if math == hip.MATH_F32 else plan.split_weight(s["w"], math)
            a["w"] = w.data_ptr()
            scale_vec = s["scale"]
            if math == hip.MATH_F16X2:  # acc = (S_in x) . (s[n] w): the power-of-two scales leave through the epilogue scale, exactly
                scale_vec = plan.descaled(s["scale"], s["w"], vin.buf.plane_scale)
            if in_planes:
                a["in_planes"] = vin.pptr
            else:
                assert vin.has_f32, f"conv {name}: the f32-input kernel reads a buffer that has split planes only"
                a["in_"] = vin.ptr
            # output forms
            wf = vout.has_f32 and s.get("write_f32", True)
            wp = bool(vout.np) and s.get("write_planes", True) and math != hip.MATH_F32
            if vout.np and wp:
                assert vout.np == hip.MATH_PLANES[math], (name, vout.np, math)
                assert not s.get("n_limit"), f"conv {name}: n_limit segments write f32 maps only"
            assert wf or wp, f"conv {name}: segment {i} writes nothing"
            a["out"] = vout.ptr if wf else 0
            a["out_planes"] = vout.pptr if wp else 0
            self.out_forms.append((wf, wp))
            a["scale"], a["bias"] = scale_vec.data_ptr(), s["bias"].data_ptr()
            a["lo"] = s["lo"].data_ptr() if s.get("lo") is not None else 0
            a["B"], a["H"], a["W"], a["Ho"], a["Wo"] = vin.B, vin.H, vin.W, Ho, Wo
            a["in_pitch"], a["out_pitch"] = vin.pitch, vout.pitch
            a["M"] = m_list[i]
            res = s.get("res")
            res_form = None
            if res is not None:
                # residual source forms (include/dd3d_hip.h, dd3d_conv_seg.res_mode): the split planes when the launch runs on the
                # split-plane kernels and the source has them (no f32 twin needed), else the f32 map; `res_up`: the source is the
                # map at half the resolution (FPN top-down: nearest x2 + add fused into the lateral convolution), planes only
                assert res.C >= meta["N"]
                planes_ok = in_planes and res.np == hip.MATH_PLANES[math] and (math != hip.MATH_F16X2 or res.buf.plane_scale == float(plan.act_scale))
                if s.get("res_up"):
                    assert planes_ok, f"conv {name}: the upsampled residual is read from split planes"
                    assert (res.B, 2 * res.H, 2 * res.W) == (vout.B, vout.H, vout.W), (name, res.H, res.W, vout.H, vout.W)
                    a["res"], a["res_pitch"], a["res_mode"] = res.pptr, 0, 3
                    res_form = "planes_up"
                else:
                    assert (res.B, res.H, res.W) == (vout.B, vout.H, vout.W)
                    if planes_ok and not (res.has_f32 and os.environ.get("DD3D_RES_F32", "0") == "1"):
                        a["res"], a["res_pitch"], a["res_mode"] = res.pptr, 0, 2
                        res_form = "planes"
                    else:
                        assert res.has_f32, f"conv {name}: the residual source has no f32 storage and its planes do not fit this launch"
                        a["res"], a["res_pitch"], a["res_mode"] = res.ptr, res.pitch, 1
                        res_form = "f32"
            self.res_forms.append(res_form)
            a["n_limit"] = int(s.get("n_limit", 0))
            assert a["n_limit"] <= meta["N"]
            tiles += [(i, m0) for m0 in range(0, m_list[i], bm)]
            self.keep += [w, scale_vec, s["bias"], s.get("lo")]  # (what the launch reads; kept alive here)
        self.desc = dict(kind="conv", segs=segs, meta=meta, stride=stride, pad=pad, relu=bool(relu), in_relu=bool(in_relu),
                         in_form="planes" if in_planes else "f32", out_forms=self.out_forms, res_forms=self.res_forms)
        self.segs_host = arr  # kept alive: the library reads the host copy at every launch (seg0_host)
        self.segs_dev = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        self.tiles_dev = torch.tensor(tiles, dtype=torch.int32).to(dev)
        # split-K: private partial-sum slab + per-tile arrival counters (private, so that independent convs may overlap)
        self.ws = self.counters = None
        if sk > 1:
            ntile = len(tiles) * -(-meta["N"] // bn)
            self.ws = torch.empty(sk * ntile * bm * bn, dtype=torch.float32, device=dev)
            self.counters = torch.zeros(ntile, dtype=torch.int32, device=dev)
        L = hip.ConvLaunch()
        L.segs, L.tiles = self.segs_dev.data_ptr(), self.tiles_dev.data_ptr()
        L.workspace = self.ws.data_ptr() if self.ws is not None else None
        L.tile_counters = self.counters.data_ptr() if self.counters is not None else None
        L.nsegs, L.ntiles = len(segs), len(tiles)
        L.KH, L.KW, L.stride, L.pad = meta["KH"], meta["KW"], stride, pad
        L.Cin, L.N, L.Kpad, L.Npad = meta["Cin"], meta["N"], meta["Kpad"], meta["Npad"]
        L.relu, L.splitk, L.math_mode, L.tile_cfg = int(relu), sk, math, cfg
        L.zero_page = plan.zero_page.data_ptr()
        L.seg0_host = self.segs_host.ctypes.data  # (host copy of every segment: one segment travels by value, all are validated by the library)
        assert not in_relu or math == hip.MATH_BF16X3
        L.in_relu = int(in_relu)
        L.in_planes = int(in_planes)
        L.out_plane_scale = float(plan.act_scale)
        L.status = plan.status.data_ptr() if plan.status is not None else None
        # underflow side of the f16x2 range guard: only launches that hand planes to a following convolution are watched
        L.amax = plan.amax_slot(name) if (math == hip.MATH_F16X2 and any(wp for _, wp in self.out_forms) and not plan.dry_run
                                          and os.environ.get("DD3D_AMAX", "1") != "0") else None  # DD3D_AMAX=0: A/B measurements only
        self.L = L
        # algorithmic MACs: every segment counts the channels it stores
        # (a segment that repeats another's products -- relu(p6) beside p6 -- is marked `algorithmic=False` and not counted)
        self.macs = sum(m * (sg.get("n_limit") or meta["N"]) for m, sg in zip(m_list, segs) if sg.get("algorithmic", True)) * meta["KH"] * meta["KW"] * meta["Cin"]
        self.info = dict(name=name, M=sum(m_list), N=meta["N"], K=meta["Kpad"], tile=(bm, bn), tile_name=hip.TILE_NAMES[cfg], splitk=sk, math=math,
                         blocks=len(tiles) * -(-meta["N"] // bn) * sk, nsegs=len(segs), in_form="planes" if in_planes else "f32")

    def __call__(self, lib, stream):
        hip.check(lib.dd3d_conv2d_igemm_f32(C.byref(self.L), stream), "conv " + self.name)


class CallOp:
    """A non-convolution launch.  `desc` says what it computes on which views (kind + operands); tools and the CPU plan emulator of
    the tests read it, the launch itself does not."""
    def __init__(self, fn, name="", desc=None):
        self.fn, self.name, self.macs, self.desc = fn, name, 0, desc

    def __call__(self, lib, stream):
        self.fn(lib, stream)


class OpList(list):
    """Launch sequence; every appended op is tagged with the branch it runs on (0 = the main stream) and with the side branches
    that must have finished before it starts (PlanBase.branch / PlanBase.join)."""
    def __init__(self, plan):
        super().__init__()
        self.plan = plan

    def append(self, op):
        op.branch = self.plan._branch
        op.joins = tuple(self.plan._pending_joins)
        self.plan._pending_joins = []
        super().append(op)


# --------------------------------------------------------------------------------------------- the plan
class PlanBase:
    """Buffer / workspace bookkeeping and op helpers shared by the full forward plan and the kernel unit tests."""
    def __init__(self, device, dry_run=False):
        self.lib = hip.lib()
        self.device = torch.device(device)
        self.dry_run = dry_run  # plan construction only (host-logic tests on a GPU-less box); launching is refused
        assert dry_run or self.device.type == "cuda", "dd3d_amd runs on an MI355X HIP device only (no CPU fallback)"
        self._branch, self._pending_joins = 0, []
        self._side_streams = {}
        import os
        # Side branches (see branch()) are OFF by default: measured A/B on one MI355X, DD3D-DLA34 B=1 graph replay 1.758 ms without
        # vs 1.783 ms with them -- the cross-stream edges cost more than the overlap of the short residual / lateral / P6-P7
        # chains returns (their neighbours already fill the CUs).  DD3D_BRANCHES=1 turns them on.
        self.use_branches = os.environ.get("DD3D_BRANCHES", "0") == "1"
        # Round 4: tensors that only convolutions, residual adds, 2x2 pools and the FPN top-down sum read exist as split planes ONLY (those
        # consumers read planes: dd3d_conv_seg.res_mode 2 / 3, dd3d_maxpool2x2_planes_in).  DD3D_PLANES_ONLY=0 keeps round 3's f32 twins
        # and the separate top-down kernels (A/B measurements).
        self.planes_only = os.environ.get("DD3D_PLANES_ONLY", "1") != "0"
        self.ops = OpList(self)
        self.bufs = {}
        self.graph = None
        self.world_size = 1
        self.zero_page = torch.zeros(64, dtype=torch.float32, device=self.device)  # padded-tap source of the DMA conv
        self.math = default_math()
        # Packed filters, their 16-bit term planes and the de-scaled epilogue vectors.  A plan built for a model shares the MODEL's store
        # (ForwardPlan.__init__ -> adopt_weight_store): every plan / pipeline slot of the model then reads the SAME device copies, so steps
        # in flight on several slots hit the same L2 / MALL lines instead of streaming one private copy of the weights per slot.
        self._packed, self._split, self._descaled = {}, {}, {}
        # filters built on the fly (pack(cache=False): grouped / re-laid filters whose storage the model does not own) keep their term planes
        # in the PLAN, so that they die with it instead of accumulating in the model's store (round-3 advisor)
        self._split_local = {}
        import math as _math
        # DD3D_MATH_F16X2: every split-plane activation holds value * act_scale (a power of two; |value| <= 65504 / act_scale or the
        # status word trips and the forward raises; terms below 2^-24 / act_scale are lost).  DD3D_F16_ACT_SCALE overrides.
        self.act_scale = float(os.environ.get("DD3D_F16_ACT_SCALE", "16"))
        assert self.act_scale > 0 and _math.log2(self.act_scale).is_integer(), "DD3D_F16_ACT_SCALE must be a power of two"
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)  # DD3D_STATUS_* bits OR-ed in by the kernels
        # DD3D_MATH_F16X2, the other side of the range guard: one float per convolution launch that writes split planes, holding the
        # largest |value * plane scale| it stored (dd3d_conv_launch.amax); zeroed at the start of a forward, read by check_status()
        self.amax = torch.zeros((512, 16, 32), dtype=torch.float32, device=self.device)  # [launch][sub-maximum][128-byte line]
        self.amax_names = []

    @property
    def use_planes(self):
        """Convolutions hand their outputs to the next convolution as split planes (csrc/conv_planes.hip) -- always in the reduced
        modes (they have no f32-input kernel); in the three-term mode unless DD3D_PLANES=0 selects the round-1 data flow (f32 NHWC
        everywhere, operands split on the fly by the consumer) for A/B measurements."""
        import os
        if self.math == hip.MATH_F32:
            return False
        return self.math != hip.MATH_BF16X3 or os.environ.get("DD3D_PLANES", "1") != "0"

    AMAX_FLOOR = 2.0**-5  # largest scaled entry of a tensor below this: > 4 of its 24 bits are under the half pair's absolute floor 2^-25

    def amax_slot(self, name):
        """Device address of a fresh per-launch maximum (DD3D_MATH_F16X2 range guard, underflow side)."""
        assert len(self.amax_names) < self.amax.shape[0]
        self.amax_names.append(name)
        return self.amax[len(self.amax_names) - 1].data_ptr()

    def amax_values(self):
        """Largest sampled |value * plane scale| of every watched launch of the last forward (CPU tensor, order of `amax_names`)."""
        return self.amax[:len(self.amax_names), :, 0].amax(1).cpu()

    HALF_MAX = 65504.0

    def range_headroom(self):
        """How close the last forward came to the two ends of the f16x2 range guard (None for the other arithmetic modes / before a
        forward): per-launch sampled maxima of |value * plane scale| against the half format's largest finite value (overflow side: the
        status bit trips per ELEMENT at 65504, so `overflow_headroom_x` < ~4 on a sample means real data may trip it) and against
        AMAX_FLOOR (underflow side, per tensor).  Reads the maxima from the device: call after the forward has been waited for."""
        if self.math != hip.MATH_F16X2 or not self.amax_names:
            return None
        mx = self.amax_values().tolist()
        seen = [(n, v) for n, v in zip(self.amax_names, mx) if v > 0.0]
        if not seen:
            return None
        hi_n, hi = max(seen, key=lambda t: t[1])
        lo_n, lo = min(seen, key=lambda t: t[1])
        return {"plane_scale": self.act_scale, "launches_watched": len(seen),
                "largest_scaled_activation": hi, "largest_in": hi_n, "overflow_headroom_x": self.HALF_MAX / hi,
                "largest_activation": hi / self.act_scale, "overflow_at": self.HALF_MAX / self.act_scale,
                "smallest_launch_maximum_scaled": lo, "smallest_in": lo_n, "underflow_headroom_x": lo / self.AMAX_FLOOR}

    def check_status(self):
        """Raise if a kernel flagged a numeric fault (reads one int32 and the per-launch maxima from the device; call after the forward
        has been waited for).  DD3D_MATH_F16X2 keeps activations as two IEEE halves of value * plane scale: exact to 2^-24 relative
        between 2^-1 and 65504, with an ABSOLUTE floor of 2^-25 below.  Overflow is flagged per element (status bit); underflow per
        tensor: a convolution whose LARGEST output, scaled, stayed below 2^-5 has lost more than four of its 24 bits."""
        st = int(self.status.cpu())
        if st & hip.STATUS_F16_OVERFLOW:
            self.status.zero_()
            raise FloatingPointError(
                f"an activation left the half range while being split (|x| > {65504.0 / self.act_scale:g} at plane scale {self.act_scale:g}, or a "
                "NaN / inf): lower DD3D_F16_ACT_SCALE or run this model with math='bf16x3'")
        if self.amax_names:
            mx = self.amax_values()
            low = [(n, float(v)) for n, v in zip(self.amax_names, mx.tolist()) if 0.0 < v < self.AMAX_FLOOR]
            if low:
                n, v = min(low, key=lambda t: t[1])
                raise FloatingPointError(
                    f"the outputs of {len(low)} convolution(s) sit below the half range's useful part (smallest: {n}, max |x| = "
                    f"{v / self.act_scale:.3g} at plane scale {self.act_scale:g}; the pair (hi, lo) has an absolute floor of {2.0**-25 / self.act_scale:.2g}): "
                    "raise DD3D_F16_ACT_SCALE or run this model with math='bf16x3'")
            # not a fault yet, but close: the sampled maximum of some launch is within DD3D_RANGE_WARN_X (default 4) of the half format's
            # largest value -- a real checkpoint's user sees how near the fallback to bf16x3 (half the throughput) is before it happens
            top = float(mx.max()) if mx.numel() else 0.0
            warn_x = float(os.environ.get("DD3D_RANGE_WARN_X", "4"))
            if top > 0.0 and self.HALF_MAX / top < warn_x and not getattr(self, "_range_warned", False):
                import warnings
                self._range_warned = True
                n = self.amax_names[int(mx.argmax())]
                warnings.warn(f"dd3d_amd: f16x2 range headroom is {self.HALF_MAX / top:.2f}x (launch {n}: sampled max |x| = {top / self.act_scale:.4g}, "
                              f"overflow at {self.HALF_MAX / self.act_scale:g}); lower DD3D_F16_ACT_SCALE or expect the bf16x3 fallback")

    def adopt_weight_store(self, model):
        """Use the model's weight store (created on first use; dropped by DD3D.invalidate_plans when the weights change)."""
        store = model.__dict__.setdefault("_weight_store", {})
        dev = store.setdefault(str(self.device), {"packed": {}, "split": {}, "descaled": {}})
        self._packed, self._split, self._descaled = dev["packed"], dev["split"], dev["descaled"]

    def pack(self, weights, cache=True):
        """pack_filter with the plan's store in front: one packed copy per filter (list of filters) and device.  `cache=False` for
        filters built on the fly (their storage is not owned by the model, so its address may be recycled)."""
        if not cache:
            wp, meta = pack_filter(weights, self.device)
            self._split_local[wp.data_ptr()] = {"wp": wp}  # (keeps the tensor alive: its address is the key)
            return wp, meta
        ws = list(weights) if isinstance(weights, (list, tuple)) else [weights]
        key = tuple((w.data_ptr(), tuple(w.shape), w._version) for w in ws)
        if key not in self._packed:
            self._packed[key] = (ws, pack_filter(weights, self.device))  # (the sources stay referenced: their addresses are the key)
        return self._packed[key][1]

    def split_weight(self, wp, math=hip.MATH_BF16X3):
        """16-bit term planes of a packed filter, built once per filter and mode (the towers share theirs over 5 levels)."""
        key = (wp.data_ptr(), math)
        store = self._split_local[wp.data_ptr()] if wp.data_ptr() in self._split_local else self._split
        if key not in store:
            if math == hip.MATH_F16X2:
                planes, row_scale = split_f16x2_host(wp)
                store[key] = (wp, planes.to(self.device), row_scale)
            else:
                store[key] = (wp, split_planes_host(wp, math).to(self.device), None)
        return store[key][1]

    def descaled(self, scale, wp, in_scale):
        """Epilogue scale of a DD3D_MATH_F16X2 convolution: scale[n] / (in_scale * row_scale[n]), all powers of two (exact).  Keyed by the
        VALUES of `scale` (every plan makes fresh device copies of the folded norms: keyed by address, each plan build added entries that
        were never freed -- round-3 advisor), so all plans / pipeline slots of a model share one vector per (norm, filter, input scale)."""
        local = wp.data_ptr() in self._split_local
        row_scale = (self._split_local[wp.data_ptr()] if local else self._split)[(wp.data_ptr(), hip.MATH_F16X2)][2]
        host = scale.detach().float().cpu().contiguous()
        key = (host.numpy().tobytes(), wp.data_ptr(), float(in_scale))  # (the bytes themselves: a few KB per vector, exact)
        store = self._split_local[wp.data_ptr()] if local else self._descaled
        if key not in store:
            n = host.numel()
            store[key] = (host / (row_scale[:n] * float(in_scale))).to(self.device)
        return store[key]

    # ------------------------------------------------------------------ helpers
    def buf(self, name, B, H, W, Cc, kind="f32"):
        """kind: which storages the tensor needs -- "f32" (read by a non-convolution kernel / as a residual / by the host), "planes"
        (read by convolutions only), "both".  Without split planes in the plan (f32 math, DD3D_PLANES=0) everything is f32."""
        assert kind in ("f32", "planes", "both"), kind
        planes = hip.MATH_PLANES[self.math] if (self.use_planes and kind != "f32" and Cc % 32 == 0) else 0
        b = Buf(B, H, W, Cc, self.device, name, f32=(kind != "planes" or not planes), planes=planes, dry_run=self.dry_run,
                f16=self.math == hip.MATH_F16X2, plane_scale=self.act_scale if self.math == hip.MATH_F16X2 else 1.0)
        self.bufs[name] = b
        return b

    def split(self, view, relu=False, dst=None, name=""):
        """f32 slice -> its split planes (the entry into the plane form for tensors a non-convolution kernel, the stem or an f32-math
        convolution wrote).  `dst`: another buffer's slice (LastLevelP6P7: the planes of relu(p6))."""
        dst = view if dst is None else dst
        assert view.has_f32 and dst.np and view.C % 32 == 0 and dst.C == view.C, (name, view.C, dst.C)
        M = view.B * view.H * view.W
        assert (dst.B, dst.H, dst.W) == (view.B, view.H, view.W)

        def _f(lib, st, view=view, dst=dst, M=M):
            hip.check(lib.dd3d_split_planes(view.ptr, dst.pptr, M, view.C, view.pitch, self.math, int(relu), dst.buf.plane_scale, self.status.data_ptr(),
                                            st), "split_planes " + name)

        self.ops.append(CallOp(_f, name or "split", dict(kind="split_planes", src=view, dst=dst, relu=bool(relu))))

    def f32_written(self, view, name=""):
        """A kernel that writes f32 only has just filled `view`: bring the buffer's split planes (if it has any) up to date."""
        if view.np:
            self.split(view, name=(name or view.buf.name) + ".split")

    def _vec(self, t):
        return t.detach().float().contiguous().to(self.device)

    def conv_module(self, conv, vin, vout, relu=False, res=None, norm=None, name="", in_relu=False, weight=None, write_f32=True, write_planes=True,
                    res_up=False):
        """One Conv2d(+folded norm)(+residual)(+relu) as a single-segment launch.  `weight`: an OIHW filter to use instead of the
        module's (the same filter re-laid for a padded input layout, see `scatter_in_channels`).  `write_f32` / `write_planes`: drop
        one of the output buffer's storages for this producer (e.g. an f32 copy nobody reads)."""
        scale, shift = fold_norm(conv, norm)
        explicit = weight is not None
        weight = dense_filter(conv) if weight is None else weight
        N, Cin, KH, KW = weight.shape
        cin_p = 4 if Cin <= 4 else 16
        # (measured in-graph: the patch kernel takes 30 / 22 us where the im2col f32 kernel took 97 / 67 on base_layer / level0;
        # on the stride-2 Cin-16 level1 the patch is 4.6 inputs per output and the im2col kernel stays 3 us ahead)
        if (not (getattr(conv, "groups", 1) > 1 or explicit) and self.math != hip.MATH_F32 and Cin <= 16 and res is None and vin.C == cin_p
                and not (cin_p == 16 and conv.stride == 2) and self.lib.dd3d_conv2d_smallc_supported(cin_p, KH, KW, conv.stride, conv.padding, N)):
            op = SmallcConvOp(self, conv.weight, cin_p, conv.stride, conv.padding, vin, vout, self._vec(scale), self._vec(shift), relu, name)
            self.ops.append(op)
            self.f32_written(vout, name)
            return op
        w, meta = self.pack(weight, cache=not explicit and getattr(conv, "groups", 1) == 1)
        seg = {"in": vin, "out": vout, "w": w, "scale": self._vec(scale), "bias": self._vec(shift), "res": res, "res_up": bool(res_up),
               "write_f32": write_f32, "write_planes": write_planes}
        op = ConvOp(self, meta, conv.stride, conv.padding, [seg], relu, name=name, in_relu=in_relu)
        self.ops.append(op)
        if op.math == hip.MATH_F32 and write_planes:
            self.f32_written(vout, name)  # an f32-math kernel (stem-sized Cin, narrow N on f32 input) writes f32 only
        return op

    def maxpool(self, vin, vout, name="pool"):
        assert vin.C == vout.C and vout.H * 2 == vin.H and vout.W * 2 == vin.W
        if not vin.has_f32:  # the map exists as split planes only: pool the planes (the winners' terms are copied)
            assert vin.np and vout.np == vin.np and not vout.has_f32 and vin.C % 32 == 0, (name, vin.np, vout.np, vout.has_f32)

            def _fq(lib, st, vin=vin, vout=vout):
                hip.check(lib.dd3d_maxpool2x2_planes_in(vin.pptr, vout.pptr, vin.B, vin.H, vin.W, vin.C, self.math, st), name)

            self.ops.append(CallOp(_fq, name, dict(kind="maxpool2x2", vin=vin, vout=vout, planes=True, in_form="planes")))
            return
        if vout.np and vout.C % 32 == 0:  # pooled map + its split planes in one launch

            def _fp(lib, st, vin=vin, vout=vout):
                hip.check(lib.dd3d_maxpool2x2_planes(vin.ptr, vout.ptr or None, vout.pptr, vin.B, vin.H, vin.W, vin.C, vin.pitch, vout.pitch, self.math,
                                                     vout.buf.plane_scale, self.status.data_ptr(), st), name)

            self.ops.append(CallOp(_fp, name, dict(kind="maxpool2x2", vin=vin, vout=vout, planes=True)))
            return

        def _f(lib, st, vin=vin, vout=vout):
            hip.check(lib.dd3d_maxpool2x2_nhwc(vin.ptr, vout.ptr, vin.B, vin.H, vin.W, vin.C, vin.pitch, vout.pitch, st), name)

        self.ops.append(CallOp(_f, name, dict(kind="maxpool2x2", vin=vin, vout=vout)))
        self.f32_written(vout, name)

    def upsample_add(self, fine, coarse, name="fpn_topdown"):
        assert fine.C == coarse.C and coarse.H * 2 == fine.H and coarse.W * 2 == fine.W
        if fine.np and fine.C % 32 == 0:  # top-down sum + its split planes in one launch

            def _fp(lib, st, fine=fine, coarse=coarse):
                hip.check(lib.dd3d_upsample2x_add_planes(fine.ptr, coarse.ptr, fine.pptr, fine.B, fine.H, fine.W, fine.C, fine.pitch, coarse.pitch, self.math,
                                                         fine.buf.plane_scale, self.status.data_ptr(), st), name)

            self.ops.append(CallOp(_fp, name, dict(kind="upsample2x_add", fine=fine, coarse=coarse, planes=True)))
            return

        def _f(lib, st, fine=fine, coarse=coarse):
            hip.check(
                lib.dd3d_upsample2x_add_nhwc(fine.ptr, coarse.ptr, fine.B, fine.H, fine.W, fine.C, fine.pitch, coarse.pitch, st), name
            )

        self.ops.append(CallOp(_f, name, dict(kind="upsample2x_add", fine=fine, coarse=coarse)))
        self.f32_written(fine, name)

    def ese(self, x, identity, out, fc, name="ese"):
        Cc, HW = x.C, x.H * x.W
        rs = max(1, min(64, HW // 256))
        cr = fc.out_channels  # real channels; the buffers may be padded to a 32-multiple (zero channels stay zero: 0 * gate + 0)
        w = torch.zeros((Cc, Cc), dtype=torch.float32)
        w[:cr, :cr] = fc.weight.detach().float().reshape(cr, cr).cpu()
        b = torch.zeros(Cc, dtype=torch.float32)
        b[:cr] = fc.bias.detach().float().cpu()
        w, b = self._vec(w), self._vec(b)
        partial = torch.zeros((x.B, rs, Cc), dtype=torch.float32, device=self.device)
        mean = torch.zeros((x.B, Cc), dtype=torch.float32, device=self.device)
        counters = torch.zeros(x.B, dtype=torch.int32, device=self.device)  # per-image arrival counters of the pooling pass; the kernel leaves them zero
        import os
        fused = os.environ.get("DD3D_ESE_FUSED", "1") != "0"  # 0: the three-launch dd3d_ese_nhwc + a separate split (A/B measurements)
        planes = fused and bool(out.np) and Cc % 32 == 0

        def _f(lib, st):
            if not fused:
                hip.check(
                    lib.dd3d_ese_nhwc(x.ptr, identity.ptr if identity is not None else None, out.ptr, w.data_ptr(), b.data_ptr(), partial.data_ptr(),
                                      mean.data_ptr(), x.B, HW, Cc, x.pitch, identity.pitch if identity is not None else 0, out.pitch, rs, st), name)
                return
            # pool (+ per-image mean), then gate + scale (+ identity) -> f32 and split planes of the result: two launches, no separate split
            hip.check(
                lib.dd3d_ese_fused(x.ptr, identity.ptr if identity is not None else None, out.ptr if out.buf.has_f32 else None,
                                   out.pptr if planes else None, w.data_ptr(), b.data_ptr(), partial.data_ptr(), mean.data_ptr(), counters.data_ptr(), x.B,
                                   HW, Cc, x.pitch, identity.pitch if identity is not None else 0, out.pitch, rs, self.math,
                                   out.buf.plane_scale if planes else 1.0, self.status.data_ptr(), st), name)

        op = CallOp(_f, name, dict(kind="ese", x=x, identity=identity, out=out, weight=w, bias=b, planes=planes))
        op.keep = [w, b, partial, mean, counters]
        self.ops.append(op)
        if out.np and not planes:
            self.f32_written(out, name)

    # ------------------------------------------------------------------ side branches
    def branch(self, b):
        """with plan.branch(b): ops appended inside run on side stream b, concurrently with what the main stream does until
        plan.join(b).  Used for short independent chains next to a long op (DLA: pool -> project beside the block's first conv;
        FPN: the other laterals beside lateral5/output5, P6/P7 beside the top-down path).  Inside a captured hipGraph the
        fork / join become graph edges."""
        plan = self

        class _Ctx:
            def __enter__(self_inner):
                self_inner.prev = plan._branch
                plan._branch = b if plan.use_branches else 0

            def __exit__(self_inner, *exc):
                plan._branch = self_inner.prev

        return _Ctx()

    def join(self, b):
        """The next op appended (on the main stream) waits for side branch b."""
        if self.use_branches:
            self._pending_joins.append(b)

    # ------------------------------------------------------------------ execution
    def launch(self, first=0, last=None):
        if self.dry_run:
            raise RuntimeError("dry-run plan: there is no CPU execution path")
        main = torch.cuda.current_stream()
        st = hip.current_stream()
        ahead = set()  # side branches holding work the main stream has not waited for yet
        for op in self.ops[first:last]:
            for j in op.joins:
                if j in ahead:
                    main.wait_stream(self._side_streams[j])
                    ahead.discard(j)
            if op.branch == 0:
                op(self.lib, st)
                continue
            side = self._side_streams.get(op.branch)
            if side is None:
                side = self._side_streams[op.branch] = torch.cuda.Stream(device=self.device)
            if op.branch not in ahead:
                side.wait_stream(main)
                ahead.add(op.branch)
            with torch.cuda.stream(side):
                op(self.lib, hip.current_stream())
        for j in ahead:
            main.wait_stream(self._side_streams[j])

    def capture(self):
        """Capture the whole launch sequence into one hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture)."""
        assert not getattr(self, "exchange", False), "graph capture is per-phase in multi-GPU mode (see dd3d_amd.parallel)"
        self.launch()  # warm-up: sets kernel attributes, faults pages
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.launch()
        self.graph = g
        return g

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.launch()

    @property
    def conv_macs(self):
        return sum(op.macs for op in self.ops)

    def describe(self):
        return [op.info for op in self.ops if isinstance(op, ConvOp)]


class ForwardPlan(PlanBase):
    """Static launch plan of DD3D.forward (inference) for one (B, Hp, Wp)."""
    def __init__(self, model, B, Hp, Wp, device=None, world_size=1, dry_run=False, rank=0, exchange=None, camera_sharded=False):
        super().__init__(device or model.device, dry_run=dry_run)
        self.camera_sharded = camera_sharded
        # the candidate exchange between select/decode and NMS exists when there are several ranks; `exchange=True` keeps its
        # buffers and the two-phase launch for one rank too (single-GPU check of the RCCL transport, tests/gpu_rccl_check.py)
        self.exchange = world_size > 1 if exchange is None else bool(exchange)
        self.adopt_weight_store(model)
        self._trunk(model, B, Hp, Wp)
        # ---- heads + post-processing
        self._heads(model, self.features)
        self._postprocess(model, world_size, rank)

    def _trunk(self, model, B, Hp, Wp):
        """Static inputs, pre-processing, backbone and FPN (shared with DenseDepthPlan)."""
        if getattr(model, "math", None) is not None:
            self.math = MATH_NAMES[model.math] if isinstance(model.math, str) else int(model.math)
        self.model = model
        self.B, self.Hp, self.Wp = B, Hp, Wp
        dev = self.device

        # ---- static inputs
        self.in_u8 = torch.zeros((B, 3, Hp, Wp), dtype=torch.uint8, device=dev)
        self.in_sizes = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        self.in_K = torch.zeros((B, 9), dtype=torch.float32, device=dev)
        self.in_outsize = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self.inv_K = torch.zeros((B, 9), dtype=torch.float32, device=dev)

        # ---- preprocess.  With the fused stem (DLA, two-half-term arithmetic: FusedStemOp) the normalised image exists only inside
        # that kernel's LDS tiles; `normalized_image()` produces it on demand (tests).
        from dd3d_amd.modeling.dla import DLA
        bb = model.backbone
        self.fused_stem = self._can_fuse_stem(bb.bottom_up) if isinstance(bb.bottom_up, DLA) else False
        self._norm = ((C.c_float * 3)(*[float(v) for v in model.pixel_mean.flatten().tolist()]),
                      (C.c_float * 3)(*[float(v) for v in model.pixel_std.flatten().tolist()]))
        img = self.buf("img4", B, Hp, Wp, 4) if (not self.fused_stem or self.dry_run) else None

        def _pre(lib, st, img=img):
            self.amax[:max(1, len(self.amax_names))].zero_()  # (captured with the rest of the forward: the maxima are per forward)
            if img is not None:
                hip.check(
                    lib.dd3d_preprocess_u8_nhwc4(self.in_u8.data_ptr(), self.in_sizes.data_ptr(), img.t.data_ptr(), B, Hp, Wp, self._norm[0], self._norm[1], st),
                    "preprocess"
                )
            hip.check(lib.dd3d_invert_intrinsics(self.in_K.data_ptr(), self.inv_K.data_ptr(), B, st), "invert_intrinsics")

        self.ops.append(CallOp(_pre, "preprocess", dict(kind="preprocess", img=img, mean=list(self._norm[0]), std=list(self._norm[1]))))

        # ---- backbone + FPN
        img_view = img.view() if img is not None else None  # (None: the fused stem reads the uint8 input itself)
        if isinstance(bb.bottom_up, DLA):
            feats = self._dla(bb.bottom_up, img_view)
        else:
            feats = self._vovnet(bb.bottom_up, img_view)
        self.bottom_up = feats
        outs = self._fpn(bb, feats)  # name -> view, finest first
        # the heads see DD3D.IN_FEATURES (core.py:32-34,84): all FPN outputs in every reference config, a subset is allowed
        self.features = [outs[n] for n in getattr(model, "in_features", list(outs))]
        if self.fpn_tail_join is not None:
            self.join(self.fpn_tail_join)  # P6 / P7 (side branch) feed the towers
        self.strides = [s.stride for s in model.backbone_output_shape]

    def normalized_image(self):
        """The padded, normalised input canvas (B, 3, Hp, Wp) as the preprocess kernel writes it -- from the plan's buffer when the plan
        has one, else (fused stem) by running that kernel into a scratch buffer on the current stream."""
        if "img4" in self.bufs:
            return self.bufs["img4"].nchw(0, 3)
        t = torch.zeros((self.B, self.Hp, self.Wp, 4), dtype=torch.float32, device=self.device)
        hip.check(self.lib.dd3d_preprocess_u8_nhwc4(self.in_u8.data_ptr(), self.in_sizes.data_ptr(), t.data_ptr(), self.B, self.Hp, self.Wp,
                                                    self._norm[0], self._norm[1], hip.current_stream()), "preprocess")
        return t[..., :3].permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ DLA-34 (dla.py:170-355)
    def _block(self, m, x, residual, out, name, join=None, out_f32=True):
        """BasicBlock (dla.py:50-62): conv1+norm+relu, conv2+norm (+residual) relu.  `join`: side branch that produces the
        residual; it runs beside conv1."""
        mid = self.buf(name + ".mid", out.B, out.H, out.W, m.conv1.out_channels, kind="planes")  # conv1 -> conv2 only
        self.conv_module(m.conv1, x, mid.view(), relu=True, name=name + ".conv1")
        if join is not None:
            self.join(join)
        self.conv_module(m.conv2, mid.view(), out, relu=True, res=residual, name=name + ".conv2", write_f32=out_f32)

    @property
    def _twin(self):
        """Storage kind of a tensor that convolutions read AND a residual add / 2x2 pool / top-down sum reads: planes only when those
        consumers read planes (planes_only), else both."""
        return "planes" if self.planes_only else "both"

    def _tree(self, m, x, name, dst=None, cat=None, bottom=None, bottom_branch=None):
        """Tree.forward (dla.py:233-247) with the root's torch.cat realised by channel placement: the root reads
        one NHWC buffer [x2 | x1 | children...] whose slices are written in place by their producers."""
        B = x.B
        Ho, Wo = x.H // m.stride, x.W // m.stride
        oc, ic = m.out_channels, m.in_channels
        if m.levels == 1:
            if cat is None:
                cat = self.buf(name + ".cat", B, Ho, Wo, m.root_dim, kind=self._twin)  # root input (planes); x1 / bottom also feed residual adds
                if m.level_root:
                    bottom = cat.view(2 * oc, ic)
                    if m.stride > 1:
                        with self.branch(1):
                            self.maxpool(x, bottom, name + ".pool")
                        bottom_branch = 1
                    else:
                        raise NotImplementedError("level_root without downsample does not occur in DLA-34")
            # downsample / project (the residual path) only meet the main path at tree1.conv2 (and at the root, later): they run
            # on side branch 1, beside tree1.conv1
            side = bottom_branch
            if bottom is None:
                if m.stride > 1:
                    bottom = self.buf(name + ".bottom", B, Ho, Wo, ic, kind=self._twin).view()
                    with self.branch(1):
                        self.maxpool(x, bottom, name + ".pool")
                    side = 1
                else:
                    bottom = x
            if m.project is not None:
                residual = self.buf(name + ".proj", B, Ho, Wo, oc, kind="planes" if self.planes_only else "f32").view()
                with self.branch(1):
                    self.conv_module(m.project, bottom, residual, name=name + ".project")
                side = 1
            else:
                residual = bottom
            x1, x2 = cat.view(oc, oc), cat.view(0, oc)
            self._block(m.tree1, x, residual, x1, name + ".tree1", join=side)
            self._block(m.tree2, x1, x1, x2, name + ".tree2", out_f32=not x2.np)  # x2 only feeds the root (planes)
            if dst is None:
                dst = self.buf(name + ".out", B, Ho, Wo, oc, kind=self._twin).view()  # next level: conv input + max-pool input
            self.conv_module(m.root.conv, cat.view(), dst, relu=True, name=name + ".root")
            return dst
        assert m.levels == 2, "DLA-34 only nests trees two deep"
        cat2 = self.buf(name + ".cat", B, Ho, Wo, m.tree2.root_dim, kind=self._twin)
        off = 2 * oc
        bottom = None
        bb = None
        if m.level_root:
            bottom = cat2.view(off, ic)
            with self.branch(1):
                self.maxpool(x, bottom, name + ".pool")
            bb = 1
            off += ic
        t1 = cat2.view(off, oc)
        self._tree(m.tree1, x, name + ".tree1", dst=t1, bottom=bottom, bottom_branch=bb)  # tree1 pools the same x: share `bottom`
        return self._tree(m.tree2, t1, name + ".tree2", dst=dst, cat=cat2)

    def _block_any(self, m, x, residual, out, name):
        """BasicBlock (dla.py:24-62) or Bottleneck (dla.py:65-100: 1x1 -> relu -> 3x3 (stride) -> relu -> 1x1, += residual, relu)."""
        from dd3d_amd.modeling.dla import Bottleneck
        if not isinstance(m, Bottleneck):
            return self._block(m, x, residual, out, name)
        c = m.conv1.out_channels
        b1 = self.buf(name + ".b1", x.B, x.H, x.W, c, kind="planes")
        self.conv_module(m.conv1, x, b1.view(), relu=True, name=name + ".conv1")
        b2 = self.buf(name + ".b2", out.B, out.H, out.W, c, kind="planes")
        self.conv_module(m.conv2, b1.view(), b2.view(), relu=True, name=name + ".conv2")
        self.conv_module(m.conv3, b2.view(), out, relu=True, res=residual, name=name + ".conv3")

    def _tree_generic(self, m, x, name, dst=None, cat=None, off=0):
        """Tree.forward (dla.py:233-247) for any depth / block / root kind (the DLA-34 trees keep their own, side-branched lowering in
        `_tree`).  The innermost root of a tree2 chain reads ONE buffer [x2 | x1 | bottom (level_root) | x1 of the enclosing trees, outermost
        first] -- the reference's `children` list -- whose slices are written in place by their producers; `off` is the next free slice."""
        B, Ho, Wo = x.B, x.H // m.stride, x.W // m.stride
        oc, ic = m.out_channels, m.in_channels
        bottom = None
        if cat is None:  # this tree starts a chain: its innermost root fixes the buffer
            inner = m
            while inner.levels > 1:
                inner = inner.tree2
            cat = self.buf(name + ".cat", B, Ho, Wo, inner.root_dim, kind=self._twin)
            off = 2 * oc
            if m.level_root:
                if m.stride == 1:
                    raise NotImplementedError("level_root without downsample does not occur in any DLA")
                bottom = cat.view(off, ic)
                self.maxpool(x, bottom, name + ".pool")
                off += ic
        if m.levels == 1:
            if bottom is None:
                if m.stride > 1:
                    bottom = self.buf(name + ".bottom", B, Ho, Wo, ic, kind=self._twin).view()
                    self.maxpool(x, bottom, name + ".pool")
                else:
                    bottom = x
            residual = bottom
            if m.project is not None:
                residual = self.buf(name + ".proj", B, Ho, Wo, oc, kind="planes" if self.planes_only else "f32").view()
                self.conv_module(m.project, bottom, residual, name=name + ".project")
            x1, x2 = cat.view(oc, oc), cat.view(0, oc)
            self._block_any(m.tree1, x, residual, x1, name + ".tree1")
            self._block_any(m.tree2, x1, x1, x2, name + ".tree2")
            if dst is None:
                dst = self.buf(name + ".out", B, Ho, Wo, oc, kind=self._twin).view()
            self.conv_module(m.root.conv, cat.view(), dst, relu=True, res=x2 if m.root.residual else None, name=name + ".root")
            return dst
        t1 = cat.view(off, oc)
        self._tree_generic(m.tree1, x, name + ".tree1", dst=t1)  # a chain of its own
        return self._tree_generic(m.tree2, t1, name + ".tree2", dst=dst, cat=cat, off=off + oc)

    def _can_fuse_stem(self, dla):
        """The one-launch stem (csrc/stem_fused.hip) covers the DLA-34 family's stem exactly: 7x7 3->16, ONE 3x3 16->16, ONE 3x3 stride-2
        16->32, all bias-free + norm + ReLU, in the two-half-term arithmetic, on an even canvas.  DD3D_FUSED_STEM=0 keeps the launch-by-
        launch lowering (A/B measurements; it is also what every other arithmetic mode uses)."""
        import os
        if self.math != hip.MATH_F16X2 or os.environ.get("DD3D_FUSED_STEM", "1") == "0" or self.Hp % 2 or self.Wp % 2:
            return False
        convs = [dla.base_layer] + list(dla.level0) + list(dla.level1)
        want = [(16, 3, 7, 1, 3), (16, 16, 3, 1, 1), (32, 16, 3, 2, 1)]
        if len(convs) != 3:
            return False
        for cv, (n, c, k, st, pd) in zip(convs, want):
            if (tuple(cv.weight.shape) != (n, c, k, k) or cv.stride != st or cv.padding != pd or getattr(cv, "groups", 1) != 1
                    or cv.bias is not None or cv.norm is None):
                return False
        return True

    def _dla(self, dla, img):
        B, H, W = self.B, self.Hp, self.Wp
        ch = dla.channels
        if self.fused_stem:
            y = self.buf("level1.0", B, H // 2, W // 2, ch[1], kind=self._twin)  # level2: conv input + max-pool input
            self.ops.append(FusedStemOp(self, self.model, [dla.base_layer, dla.level0[0], dla.level1[0]], y.view(), name="stem"))
            x = y.view()
        else:
            base = self.buf("base", B, H, W, ch[0])
            self.conv_module(dla.base_layer, img, base.view(), relu=True, name="base_layer")
            x = base.view()
            for i, conv in enumerate(dla.level0):
                y = self.buf(f"level0.{i}", B, H, W, ch[0])
                self.conv_module(conv, x, y.view(), relu=True, name=f"level0.{i}")
                x = y.view()
            for i, conv in enumerate(dla.level1):
                y = self.buf(f"level1.{i}", B, x.H // conv.stride, x.W // conv.stride, ch[1], kind="both")  # level2: conv input + max-pool input
                self.conv_module(conv, x, y.view(), relu=True, name=f"level1.{i}")
                x = y.view()
        outs = {"level0": None, "level1": x}
        from dd3d_amd.modeling.dla import BasicBlock
        plain34 = dla.block is BasicBlock and max(dla.levels) <= 2 and not dla.residual_root  # DLA-34: the measured lowering
        for lvl in range(2, 6):
            tree = getattr(dla, f"level{lvl}")
            x = self._tree(tree, x, f"level{lvl}") if plain34 and not getattr(self.model, "force_generic_dla", False) else self._tree_generic(tree, x, f"level{lvl}")
            outs[f"level{lvl}"] = x
        return {k: outs[k] for k in dla._out_features}

    # ------------------------------------------------------------------ VoVNet-V2 (vovnet.py:218-238,357-367)
    def _vovnet(self, vov, img):
        """OSA modules with the torch.cat realised by channel placement: each module owns one NHWC buffer
        [x | layer0 | ... | layer4]; its input slice is written in place by the producer (stem conv, stage max-pool or the
        previous module's eSE + identity kernel)."""
        from dd3d_amd.modeling.vovnet import seq_conv, seq_dw, seq_norm
        B = img.B

        def conv_norm_relu(dw, conv, norm, src, dst, name, stride=1, write_f32=True):
            """conv3x3 / conv1x1 (vovnet.py:124-161), or dw_conv3x3 (:99-121): depthwise 3x3 (no norm, no relu) into a scratch buffer, then
            the pointwise 1x1 + norm + relu."""
            if dw is not None:
                Ho, Wo = (src.H + 2 - 3) // dw.stride + 1, (src.W + 2 - 3) // dw.stride + 1
                tmp = self.buf(name + ".dw", B, Ho, Wo, pad32(dw.out_channels), kind="planes").view()
                self.conv_module(dw, src, tmp, relu=False, name=name + ".dw")
                src = tmp
            self.conv_module(conv, src, dst, relu=True, norm=norm, name=name, write_f32=write_f32)

        x = img
        stages = [getattr(vov, n) for n in vov.stage_names]
        mods0 = list(stages[0].children())

        def cat_width(m):  # every slice of the concat buffer starts on a 32-channel boundary (no-op for the 32-multiple specs)
            return pad32(m.in_ch) + len(m.layers) * pad32(m.stage_ch)

        cat = None
        for idx, (cname, nname, dwname) in enumerate(vov.stem_seqs):
            conv, norm = getattr(vov.stem, cname), getattr(vov.stem, nname)
            dw = getattr(vov.stem, dwname) if dwname else None
            strided = dw if dw is not None else conv
            Ho, Wo = (x.H + 2 - 3) // strided.stride + 1, (x.W + 2 - 3) // strided.stride + 1
            if idx == len(vov.stem_seqs) - 1:
                cat = self.buf("stage2.OSA2_1.cat", B, Ho, Wo, cat_width(mods0[0]), kind="both")
                y = cat.view(0, pad32(conv.out_channels))
            else:  # stem_1 comes out of the patch kernel as f32 (split afterwards); stem_2 only feeds stem_3
                y = self.buf(f"stem.{idx}", B, Ho, Wo, conv.out_channels, kind="both" if idx == 0 else "planes").view()
            conv_norm_relu(dw, conv, norm, x, y, cname)
            x = y
        outs, prev = {}, None
        for si, (sname, stage) in enumerate(zip(vov.stage_names, stages)):
            mods = [(n, m) for n, m in stage.named_children()]
            if stage.has_pool:
                Hp = -(-(prev.H - 3) // 2) + 1
                Wp = -(-(prev.W - 3) // 2) + 1
                Hp -= (Hp - 1) * 2 >= prev.H
                Wp -= (Wp - 1) * 2 >= prev.W
                cat = self.buf(f"{sname}.{mods[0][0]}.cat", B, Hp, Wp, cat_width(mods[0][1]), kind="both")
                dstv = cat.view(0, pad32(mods[0][1].in_ch))

                def _pool(lib, st, vin=prev, vout=dstv):
                    hip.check(lib.dd3d_maxpool3x3s2_ceil_nhwc(vin.ptr, vout.ptr, vin.B, vin.H, vin.W, vin.C, vin.pitch, vout.pitch, st), "pool3")

                self.ops.append(CallOp(_pool, f"{sname}.pool", dict(kind="maxpool3x3s2_ceil", vin=prev, vout=dstv)))
                self.f32_written(dstv, f"{sname}.pool")
            H, W = cat.H, cat.W
            for k, (mname, m) in enumerate(mods):
                pin, pst, pcc = pad32(m.in_ch), pad32(m.stage_ch), pad32(m.concat_ch)
                src = cat.view(0, pin)
                if m.conv_reduction is not None:  # depthwise modules: 1x1 to stage_ch first (vovnet.py:201-205,224-225); not part of the concat
                    red = self.buf(f"{sname}.{mname}.red", B, H, W, pst, kind="planes").view()
                    self.conv_module(seq_conv(m.conv_reduction), src, red, relu=True, norm=seq_norm(m.conv_reduction), name=f"{mname}.reduction")
                    src = red
                for i, layer in enumerate(m.layers):
                    dst = cat.view(pin + i * pst, pst)
                    conv_norm_relu(seq_dw(layer), seq_conv(layer), seq_norm(layer), src, dst, f"{mname}.{i}", write_f32=not dst.np)
                    src = dst
                xt = self.buf(f"{sname}.{mname}.xt", B, H, W, pcc).view()
                segments = [(m.in_ch, pin)] + [(m.stage_ch, pst)] * len(m.layers)
                w_cat = None if all(r == q for r, q in segments) else scatter_in_channels(seq_conv(m.concat).weight.detach(), segments)
                self.conv_module(seq_conv(m.concat), cat.view(), xt, relu=True, norm=seq_norm(m.concat), name=f"{mname}.concat", weight=w_cat)
                if k + 1 < len(mods):
                    nxt = self.buf(f"{sname}.{mods[k + 1][0]}.cat", B, H, W, cat_width(mods[k + 1][1]), kind="both")
                    dst = nxt.view(0, pcc)
                else:
                    nxt = None
                    dst = self.buf(f"{sname}.out", B, H, W, pcc, kind="both").view()  # stage output: FPN lateral + next stage's pool
                self.ese(xt, cat.view(0, pin) if m.identity else None, dst, m.ese.fc, name=f"{mname}.ese")
                cat = nxt
            outs[sname] = prev = dst
        return {k: outs[k] for k in vov._out_features}

    # ------------------------------------------------------------------ FPN ([ext] detectron2 FPN.forward)
    def _fpn(self, fpn, feats):
        names = fpn.in_features
        results = {}
        # pyramid outputs feed convolutions only (towers, P6); DD3D_KEEP_F32=1 keeps f32 copies too (debugging)
        import os
        p_kind = "both" if os.environ.get("DD3D_KEEP_F32", "0") == "1" else "planes"
        assert fpn._fuse_type == "sum", "FUSE_TYPE avg is not used by any reference config"
        fused = self.planes_only and self.use_planes  # top-down sum inside the lateral convolution's epilogue (res_mode 3)
        lats = {}
        if fused:
            # [ext d2 FPN.forward]: prev = lateral(f) + interpolate(prev, x2, nearest); out = output_conv(prev) -- coarsest level first.  The
            # lateral convolution of a finer level reads the coarser level's SUM out of its split planes (pixel (h/2, w/2)) and adds it in
            # its epilogue: no f32 twin of the laterals, no fpn_topdown launches.
            # The output convolutions (3x3, Cout -> Cout on every level) only read their own lateral: ONE multi-segment launch for all
            # levels after the lateral chain (like a tower layer) instead of a launch per level -- the coarse levels' few tiles fill the
            # tail of the fine level's grid.  P6 / P7 follow (they read the coarsest output).
            prev = None
            out_segs, out_meta = [], None
            for idx in range(len(names)):
                f = feats[names[-idx - 1]]
                st = fpn.stages[-idx - 1]
                lat = self.buf(f"fpn_lateral{st}", f.B, f.H, f.W, fpn._out_feature_channels[f"p{st}"], kind="planes").view()
                lats[st] = lat
                up_ok = prev is not None and (2 * prev.H, 2 * prev.W) == (lat.H, lat.W)
                assert prev is None or up_ok, "FPN levels whose sizes are not exact halves do not occur on a size-divisible canvas"
                self.conv_module(getattr(fpn, f"fpn_lateral{st}"), f, lat, name=f"fpn_lateral{st}", res=prev, res_up=prev is not None)
                out = self.buf(f"p{st}", f.B, f.H, f.W, lat.C, kind=p_kind).view()
                conv = getattr(fpn, f"fpn_output{st}")
                scale, shift = fold_norm(conv, None)
                w, meta = self.pack(dense_filter(conv))
                assert out_meta is None or {k: meta[k] for k in ("N", "Cin", "KH", "KW", "Kpad")} == {k: out_meta[k] for k in ("N", "Cin", "KH", "KW", "Kpad")}
                assert (conv.stride, conv.padding) == (1, 1)
                out_meta = meta
                out_segs.append({"in": lat, "out": out, "w": w, "scale": self._vec(scale), "bias": self._vec(shift)})
                results[f"p{st}"] = out
                prev = lat
            self.ops.append(ConvOp(self, out_meta, 1, 1, out_segs, relu=False, name="fpn_outputs"))
            self._top_block(fpn, results, p_kind)
            self.fpn_tail_join = 3 if fpn.top_block is not None else None
            return OrderedDict((n, results[n]) for n in fpn._out_features)
        # ---- round-3 lowering (f32 math, DD3D_PLANES=0, DD3D_PLANES_ONLY=0): laterals as f32 (+ planes), separate top-down launches
        # The laterals of the finer levels only need backbone features: side branch 2, beside lateral/output of the coarsest
        # level; P6/P7 only need the coarsest output: side branch 3, beside the rest of the top-down path.
        for idx in list(range(1, len(names))) + [0]:  # side-branch ops first: a branch forks where its first op sits in the list
            f = feats[names[-idx - 1]]
            st = fpn.stages[-idx - 1]
            # laterals: f32 for the top-down sum, planes for the output conv
            lat = self.buf(f"fpn_lateral{st}", f.B, f.H, f.W, fpn._out_feature_channels[f"p{st}"], kind="both").view()
            lats[st] = lat
            if idx == 0:
                self.conv_module(getattr(fpn, f"fpn_lateral{st}"), f, lat, name=f"fpn_lateral{st}")
                out = self.buf(f"p{st}", f.B, f.H, f.W, lat.C, kind=p_kind).view()
                self.conv_module(getattr(fpn, f"fpn_output{st}"), lat, out, name=f"fpn_output{st}")
                results[f"p{st}"] = out
            else:
                with self.branch(2):  # (its planes are written after the top-down sum)
                    self.conv_module(getattr(fpn, f"fpn_lateral{st}"), f, lat, name=f"fpn_lateral{st}", write_planes=False)
        self._top_block(fpn, results, p_kind)
        prev = lats[fpn.stages[-1]]
        for idx in range(1, len(names)):
            f = feats[names[-idx - 1]]
            st = fpn.stages[-idx - 1]
            lat = lats[st]
            if idx == 1:
                self.join(2)
            self.upsample_add(lat, prev, f"fpn_topdown{st}")
            prev = lat
            out = self.buf(f"p{st}", f.B, f.H, f.W, lat.C, kind=p_kind).view()
            self.conv_module(getattr(fpn, f"fpn_output{st}"), lat, out, name=f"fpn_output{st}")
            results[f"p{st}"] = out
        self.fpn_tail_join = 3 if fpn.top_block is not None else None
        return OrderedDict((n, results[n]) for n in fpn._out_features)

    def _top_block(self, fpn, results, p_kind):
        """LastLevelP6P7 / LastLevelP6 [ext; built at dla.py:550-557]: p6 = conv(p5), p7 = conv(relu(p6)), on side branch 3."""
        if fpn.top_block is None:
            return
        st = fpn.stages[-1]
        x = results[f"p{st}"]  # in_feature "p5" is an FPN output (dla.py:550-557)
        two = fpn.top_block.num_levels == 2
        Ho, Wo = (x.H + 1) // 2, (x.W + 1) // 2
        if two and self.planes_only and self.use_planes and x.np:
            # ONE launch, two segments on the same input and filter: p6 (what the towers read) and relu(p6) (what the p7 convolution
            # reads; per-channel lower clamp 0) -- both as planes, no f32 twin of p6 and no separate split launch
            conv = fpn.top_block.p6
            p6 = self.buf(f"p{st + 1}", x.B, Ho, Wo, x.C, kind=p_kind).view()
            p6r = self.buf(f"p{st + 1}.relu", x.B, Ho, Wo, x.C, kind="planes").view()
            scale, shift = fold_norm(conv, None)
            w, meta = self.pack(dense_filter(conv))
            segs = [{"in": x, "out": o, "w": w, "scale": self._vec(scale), "bias": self._vec(shift), "lo": lo, "algorithmic": lo is None}
                    for o, lo in ((p6, None), (p6r, self._vec(torch.zeros(conv.out_channels))))]
            with self.branch(3):
                self.ops.append(ConvOp(self, meta, conv.stride, conv.padding, segs, relu=False, name="top_block.p6"))
            results[f"p{st + 1}"] = p6
            p7 = self.buf(f"p{st + 2}", x.B, (Ho + 1) // 2, (Wo + 1) // 2, x.C, kind=p_kind).view()
            with self.branch(3):
                self.conv_module(fpn.top_block.p7, p6r, p7, name="top_block.p7")
            results[f"p{st + 2}"] = p7
            return
        p6 = self.buf(f"p{st + 1}", x.B, Ho, Wo, x.C, kind="both" if two else p_kind).view()
        with self.branch(3):
            self.conv_module(fpn.top_block.p6, x, p6, name="top_block.p6")
        results[f"p{st + 1}"] = p6
        if two:
            p7 = self.buf(f"p{st + 2}", p6.B, (p6.H + 1) // 2, (p6.W + 1) // 2, p6.C, kind=p_kind).view()
            with self.branch(3):
                if p6.np:
                    # p7 = conv(relu(p6)) [ext LastLevelP6P7]: the planes of relu(p6), split from its f32 copy
                    p6r = self.buf(f"p{st + 1}.relu", p6.B, p6.H, p6.W, p6.C, kind="planes").view()
                    self.split(p6, relu=True, dst=p6r, name="top_block.p6.relu")
                    self.conv_module(fpn.top_block.p7, p6r, p7, name="top_block.p7")
                elif self.math == hip.MATH_BF16X3:
                    # p7 = conv(relu(p6)) [ext LastLevelP6P7]: the conv rectifies its input while splitting it
                    self.conv_module(fpn.top_block.p7, p6, p7, name="top_block.p7", in_relu=True)
                else:  # f32-MFMA mode: a rectified copy of p6 from a second run of its conv
                    p6r = self.buf(f"p{st + 1}.relu", p6.B, p6.H, p6.W, p6.C).view()
                    self.conv_module(fpn.top_block.p6, x, p6r, relu=True, name="top_block.p6.relu")
                    self.conv_module(fpn.top_block.p7, p6r, p7, name="top_block.p7")
            results[f"p{st + 2}"] = p7

    # ------------------------------------------------------------------ heads (fcos2d.py:130-156, fcos3d.py:160-188)
    def _heads(self, model, feats):
        dev = self.device
        h2, h3 = model.fcos2d_head, (None if model.only_box2d else model.fcos3d_head)
        L = len(feats)
        towers = [("cls", h2.cls_tower), ("box2d", h2.box2d_tower)] + ([("box3d", h3.box3d_tower)] if h3 is not None else [])
        nt = len(towers)
        Cf = feats[0].C
        depth = max(len(t) for _, t in towers)
        ping = [self.buf(f"towerA.{l}", f.B, f.H, f.W, nt * Cf, kind="planes") for l, f in enumerate(feats)]  # conv -> conv only
        pong = [self.buf(f"towerB.{l}", f.B, f.H, f.W, nt * Cf, kind="planes") for l, f in enumerate(feats)]
        cur = [[feats[l] for _ in range(nt)] for l in range(L)]  # current input view per (level, tower)
        for i in range(depth):
            dstbufs = ping if i % 2 == 0 else pong
            segs, meta = [], None
            for t, (tname, tower) in enumerate(towers):
                if i >= len(tower):
                    continue
                conv = tower[i]
                w, meta = self.pack(conv.weight)
                for l in range(L):
                    # ModuleListDial: level l uses norm[l] (normalization.py:30-40)
                    norm = conv.norm[l] if isinstance(conv.norm, torch.nn.ModuleList) else conv.norm
                    scale, shift = fold_norm(conv, norm)
                    out = dstbufs[l].view(t * Cf, Cf)
                    segs.append({"in": cur[l][t], "out": out, "w": w, "scale": self._vec(scale), "bias": self._vec(shift)})
                    cur[l][t] = out
            self.ops.append(ConvOp(self, meta, 1, 1, segs, relu=True, name=f"towers.{i}"))
        self.tower_out = cur

        C_ = model.num_classes

        pred_groups = []  # every predictor group becomes a set of segments of ONE launch (see the end of this method)

        def fused_predictor(name, convs, tower_idx, level_scale, level_bias_extra, lo):
            """convs: list of (module per level-or-shared) concatenated along N.  level_scale(l) -> per-channel scale vector."""
            ws, metas = {}, None
            n_total = sum(c[0].out_channels for c in convs)
            pitch = (n_total + 3) // 4 * 4
            segs, maps = [], []
            for l in range(L):
                key = tuple(id(c[l if len(c) > 1 else 0]) for c in convs)
                if key not in ws:
                    mods = [c[l if len(c) > 1 else 0] for c in convs]
                    w, metas = self.pack([m.weight for m in mods])
                    b = torch.cat([
                        m.bias.detach().float().cpu() if m.bias is not None else torch.zeros(m.out_channels) for m in mods
                    ])
                    ws[key] = (w, b)
                w, b = ws[key]
                sc = level_scale(l)
                bias = b * sc + level_bias_extra(l)  # (conv + b) * scale + offset, cf. fcos2d.py:146-150, fcos3d.py:175-180
                f = feats[l]
                out = self.buf(f"{name}.{l}", f.B, f.H, f.W, pitch)
                maps.append(out)
                segs.append({
                    "in": cur[l][tower_idx], "out": out.view(0, pitch), "w": w, "scale": self._vec(sc), "bias": self._vec(bias),
                    "lo": None if lo is None else self._vec(lo), "n_limit": n_total
                })
            pred_groups.append((name, metas, segs))
            return maps, pitch

        ones = lambda n: torch.ones(n)
        zeros = lambda n: torch.zeros(n)
        # cls logits (+ nuScenes attr/speed on the cls tower, nuscenes_dd3d.py:371-374)
        cls_convs = [[h2.cls_logits]]
        n_cls_extra = 0
        if hasattr(model, "attr_logits"):
            cls_convs += [[model.attr_logits], [model.speed]]
            n_cls_extra = model.attr_logits.out_channels + model.speed.out_channels
        n_cls = C_ + n_cls_extra
        lo_cls = None
        if n_cls_extra:
            lo_cls = torch.full((n_cls, ), -float("inf"))
            lo_cls[-1] = 0.0  # speed = relu(conv)
        self.cls_maps, self.cls_pitch = fused_predictor("cls_map", cls_convs, 0, lambda l: ones(n_cls), lambda l: zeros(n_cls), lo_cls)

        # box2d_reg (scale_l, relu) + centerness  (fcos2d.py:143-152)
        def s2(l):
            s = h2.scales_box2d_reg[l].scale.detach().float().cpu() if h2.use_scale else torch.ones(1)
            return torch.cat([s.expand(4), torch.ones(1)])

        lo2 = torch.tensor([0., 0., 0., 0., -float("inf")])
        self.b2d_maps, self.b2d_pitch = fused_predictor(
            "box2d_map", [[h2.box2d_reg], [h2.centerness]], 1, s2, lambda l: zeros(5), lo2
        )

        self.b3d_maps, self.b3d_pitch = None, 0
        if h3 is not None:
            C3 = 1 if h3.class_agnostic else C_

            def s3(l):
                if not h3.use_scale:
                    return ones(11 * C3)
                g = lambda ml: ml[l].scale.detach().float().cpu()
                return torch.cat([
                    ones(4 * C3), g(h3.scales_proj_ctr).expand(2 * C3), g(h3.scales_depth).expand(C3), g(h3.scales_size).expand(3 * C3),
                    g(h3.scales_conf).expand(C3)
                ])

            def b3(l):
                o = zeros(11 * C3)
                if h3.use_scale:
                    o[6 * C3:7 * C3] = h3.offsets_depth[l].bias.detach().float().cpu()
                return o

            preds = [list(h3.box3d_quat), list(h3.box3d_ctr), list(h3.box3d_depth), list(h3.box3d_size), list(h3.box3d_conf)]
            self.b3d_maps, self.b3d_pitch = fused_predictor("box3d_map", preds, 2, s3, b3, None)

        # Predictor launches.  Round 3: ONE launch, the narrow groups (C or 5 channels) riding along with the widest (11 * C) as extra
        # segments zero-padded to its Npad -- 192 executed output columns for 63 useful ones, and a kernel form that skipped the unstored
        # column blocks changed nothing (the idle waves still sat behind the block's barriers).  Round 4: the groups of <= 32 channels
        # (cls logits (+ nuScenes attr / speed), box2d + centerness) run on the 32-column tile DD3D_TILE_128x32_W4 in their own launch --
        # blocks for columns nobody stores are never created -- and the wide group(s) keep the measured tile.  DD3D_PRED_SPLIT=0: round 3's form.
        split = os.environ.get("DD3D_PRED_SPLIT", "1") != "0" and self.use_planes
        narrow = [g for g in pred_groups if split and g[1]["N"] <= 32]
        wide = [g for g in pred_groups if g not in narrow]
        for groups, tile, name in ((narrow, hip.TILE_128x32_W4, "predictors.narrow"), (wide, None, "predictors")):
            if not groups:
                continue
            n_max = max(m["N"] for _, m, _ in groups)
            npad = (n_max + 31) // 32 * 32
            meta = dict(groups[0][1], N=n_max, Npad=npad)
            all_segs = []
            for _, m, segs in groups:
                assert (m["Cin"], m["KH"], m["KW"], m["Kpad"]) == (meta["Cin"], meta["KH"], meta["KW"], meta["Kpad"])
                for sg in segs:
                    key = ("predictor_pad", sg["w"].data_ptr(), npad)
                    if key not in self._packed:  # (the store keeps the source referenced: its address is the key)
                        wpad = torch.zeros((npad, m["Kpad"]), dtype=torch.float32, device=dev)
                        wpad[:sg["w"].shape[0]] = sg["w"]
                        self._packed[key] = (sg["w"], wpad)
                    all_segs.append(dict(sg, w=self._packed[key][1]))
            self.ops.append(ConvOp(self, meta, 1, 1, all_segs, relu=False, name=name, tile=tile if npad == 32 else None))

    # ------------------------------------------------------------------ selection / decode / NMS
    def _postprocess(self, model, world_size, rank=0):
        cfg, dev, B = model.cfg, self.device, self.B
        L = len(self.features)
        inf2 = cfg.DD3D.FCOS2D.INFERENCE
        topk = int(inf2.PRE_NMS_TOPK)
        C_ = model.num_classes
        # candidate slots per image: level l can never hold more than H*W*C candidates, so it gets min(topk, H*W*C) slots -- the
        # buffer the ranks exchange carries no slot that cannot be filled (KITTI 384x1280: 3750 instead of 5000 slots)
        caps = [min(topk, f.H * f.W * C_) for f in self.features]
        self.slot_off = [sum(caps[:l]) for l in range(L + 1)]
        NS = self.slot_off[L]
        self.topk, self.num_levels, self.slots_per_image = topk, L, NS
        a = hip.SelectArgs()
        sizes = []
        for l, f in enumerate(self.features):
            a.cls[l] = self.cls_maps[l].t.data_ptr()
            a.box2d[l] = self.b2d_maps[l].t.data_ptr()
            a.box3d[l] = self.b3d_maps[l].t.data_ptr() if self.b3d_maps is not None else None
            a.H[l], a.W[l], a.stride[l] = f.H, f.W, self.strides[l]
            sizes.append(f.H * f.W * C_)
        a.cls_pitch, a.b2d_pitch, a.b3d_pitch = self.cls_pitch, self.b2d_pitch, self.b3d_pitch
        a.num_levels, a.B, a.num_classes = L, B, C_
        a.loc_offset_half = int(cfg.DD3D.FEATURE_LOCATIONS_OFFSET == "half")
        a.thresh_with_ctr = int(bool(inf2.THRESH_WITH_CTR))
        a.topk, a.pre_nms_thresh = topk, float(inf2.PRE_NMS_THRESH)
        a.attr_off, a.num_attr, a.speed_off = 0, 0, -1
        if hasattr(model, "attr_logits"):  # nuScenes extras ride on the cls map (see _heads)
            a.attr_off, a.num_attr = C_, model.attr_logits.out_channels
            a.speed_off = C_ + model.attr_logits.out_channels
        if self.b3d_maps is not None:
            c3 = cfg.DD3D.FCOS3D
            a.class_agnostic_3d = int(bool(c3.CLASS_AGNOSTIC_BOX3D))
            a.min_depth, a.max_depth = float(c3.MIN_DEPTH), float(c3.MAX_DEPTH)
            a.focal_factor = float(c3.SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR)
            a.scale_depth_by_focal = int(bool(c3.SCALE_DEPTH_BY_FOCAL_LENGTHS))
            a.allocentric = int(bool(c3.PREDICT_ALLOCENTRIC_ROT))
            a.depth_is_distance = int(bool(c3.PREDICT_DISTANCE))
            self.canon = torch.tensor([list(r) for r in c3.CANONICAL_BOX3D_SIZES], dtype=torch.float32, device=dev)
            a.canon_sizes = self.canon.data_ptr()
        a.inv_K = self.inv_K.data_ptr()
        off = 0
        for l in range(L):
            a.scratch_off[l] = off
            off += sizes[l]
        a.scratch_img_stride = off
        for l in range(L + 1):
            a.slot_off[l] = self.slot_off[l]
        self.scratch_idx = torch.empty(B * off, dtype=torch.int32, device=dev)
        self.scratch_score = torch.empty(B * off, dtype=torch.float32, device=dev)
        # What a rank hands to the others is ONE contiguous record of 4-byte words:
        #   [candidates B x F x NS | counts B x L | resize targets B x 4 | K^-1 B x 9 | camera->global pose B x 7]
        # (the last two are what the BEV stages need of an image beside its detections: with them in the record, the owner of a nuScenes
        # sample can aggregate cameras that OTHER ranks decoded).  The post-select stages read records out of the gathered buffer, so
        # they are ordered behind the collective.
        pad4 = lambda n: (n + 3) // 4 * 4
        n_c, n_k, n_o, n_i, n_p = pad4(B * hip.CAND_FIELDS * NS), pad4(B * L), pad4(B * 4), pad4(B * 9), pad4(B * 7)
        self.record_fields = dict(cand=(0, hip.CAND_FIELDS * NS), counts=(n_c, L), outsize=(n_c + n_k, 4), inv_K=(n_c + n_k + n_o, 9),
                                  pose=(n_c + n_k + n_o + n_i, 7))  # name -> (word offset in a record, words per image)
        self.flags_off = n_c + n_k + n_o + n_i + n_p  # 4 words per RECORD (not per image): the rank's range-guard verdict (below)
        self.record_len = self.flags_off + 4

        def views(rec):
            f = self.record_fields
            cut = lambda name: rec[f[name][0]:f[name][0] + B * f[name][1]]
            return (cut("cand").view(B, hip.CAND_FIELDS, NS), cut("counts").view(torch.int32).view(B, L), cut("outsize").view(B, 4),
                    cut("inv_K").view(B, 9), cut("pose").view(B, 7))

        self.record = torch.zeros(self.record_len, dtype=torch.float32, device=dev)
        self.cand, self.counts, outsize, inv_K, pose = views(self.record)
        outsize.copy_(self.in_outsize)
        self.in_outsize = outsize  # stage_inputs writes the resize targets straight into the record
        self.inv_K = inv_K         # dd3d_invert_intrinsics writes K^-1 straight into the record (the launch reads self.inv_K when it runs)
        self.in_pose = pose
        self.in_pose[:, 0] = 1.0   # identity rotation until stage_inputs fills it (models without BEV stages never do)
        self.npass = torch.zeros((B, L), dtype=torch.int32, device=dev)
        a.inv_K = self.inv_K.data_ptr()
        a.scratch_idx, a.scratch_score = self.scratch_idx.data_ptr(), self.scratch_score.data_ptr()
        a.cand, a.counts, a.npass = self.cand.data_ptr(), self.counts.data_ptr(), self.npass.data_ptr()
        self.select_args = a
        self.ops.append(CallOp(lambda lib, st: hip.check(lib.dd3d_fcos_select_decode(C.byref(a), st), "select_decode"), "select_decode"))
        if self.exchange and self.math == hip.MATH_F16X2 and not self.dry_run:
            # the rank's range-guard verdict (status bits, underflow flag) rides in its record: after the all_gather every rank sees every
            # rank's and all of them raise / fall back on the SAME step (a rank that raised alone would leave its peers in the next collective)
            flags = self.record[self.flags_off:self.flags_off + 4]

            def _flags(lib, st, flags=flags):
                hip.check(lib.dd3d_fold_range_flags(self.status.data_ptr(), self.amax.data_ptr(), len(self.amax_names), float(self.AMAX_FLOOR),
                                                    flags.data_ptr(), st), "fold_range_flags")

            self.ops.append(CallOp(_flags, "range_flags", dict(kind="range_flags")))
        self.num_pre_nms_ops = len(self.ops)

        # The exchange (dd3d_amd.parallel): every rank's record is all-gathered into `gathered` [W x record]; each rank then finalises
        # the images it OWNS out of the gathered buffer (no rank repeats another rank's NMS):
        #   * default -- its own B images, i.e. ITS segment (class-aware NMS is per image; a nuScenes sample's cameras are rank-local
        #     when the caller shards whole samples, as the reference's InferenceGroupSampler does, group_sampler.py:30-35);
        #   * camera_sharded (NuscenesDD3D, "images shard one-per-GPU"): global image g = rank * B + b, the 6 consecutive global images
        #     6 s .. 6 s + 5 are the cameras of sample s, and the rank that decoded a sample's FIRST camera owns the sample: it runs the
        #     2D NMS of all six cameras and the sample-level BEV aggregation on records other ranks delivered.
        inf = cfg.DD3D.INFERENCE
        bev_single = bool(inf.DO_BEV_NMS) and self.b3d_maps is not None
        bev_sample = bool(getattr(model, "aggregates_samples", False)) and bool(inf.DO_POSTPROCESS) and self.b3d_maps is not None
        self.world_size, self.rank = world_size, rank
        self.camera_sharded = bool(self.camera_sharded)
        if self.camera_sharded:
            ncam = int(getattr(model, "num_images_per_sample", 6))
            if not (self.exchange and bev_sample):
                raise ValueError("camera_sharded needs the candidate exchange and a model that aggregates samples (NuscenesDD3D with DO_POSTPROCESS)")
            if (world_size * B) % ncam:
                raise ValueError(f"camera_sharded: {world_size} ranks x {B} images per step do not make whole {ncam}-camera samples")
            own = [s_ for s_ in range(world_size * B // ncam) if (s_ * ncam) // B == rank]  # contiguous: the owner grows with the sample
            self.own_samples = own
            self.G = G = ncam * len(own)
            self.img_first = ncam * own[0] if own else 0
        else:
            self.own_samples = None
            self.G = G = B
            self.img_first = rank * B if self.exchange else 0
        if self.exchange:
            self.gathered = torch.zeros(world_size * self.record_len, dtype=torch.float32, device=dev)
            self.cand_all, self.counts_all, self.outsize_all, _, _ = views(self.gathered[rank * self.record_len:(rank + 1) * self.record_len])
            src = self.gathered
        else:
            self.gathered = None
            self.cand_all, self.counts_all, self.outsize_all = self.cand, self.counts, self.in_outsize
            src = self.record

        def field_ptr(name):  # record 0's block of a field in the buffer the post stages read
            return src.data_ptr() + 4 * self.record_fields[name][0]

        def addressing(args):
            args.img_first, args.img_per_rec, args.rec_stride = (self.img_first, B, self.record_len) if self.exchange else (0, 0, 0)

        self.has_bev_inputs = self.has_global_boxes = False
        self.det_cap = NS if (not inf.DO_NMS or inf2.POST_NMS_TOPK <= 0) else min(NS, int(inf2.POST_NMS_TOPK) + 156)
        self.det = torch.zeros((G, self.det_cap, hip.DET_FIELDS), dtype=torch.float32, device=dev)
        self.det_count = torch.zeros((G, ), dtype=torch.int32, device=dev)
        if bev_single or bev_sample:
            self.has_bev_inputs = True
            self.in_group = torch.zeros((B, ), dtype=torch.int32, device=dev)
        if G == 0:
            return  # a camera-sharded rank that owns no sample of the step: it only contributes its record
        ncap = (NS + 63) // 64 * 64
        n = hip.NmsArgs()
        self.sort_idx = torch.zeros((G, ncap), dtype=torch.int32, device=dev)
        self.sbox = torch.zeros((G, ncap, 4), dtype=torch.float32, device=dev)
        self.scls = torch.zeros((G, ncap), dtype=torch.int32, device=dev)
        self.mask = torch.zeros((G, ncap, ncap // 64), dtype=torch.int64, device=dev)
        self.nvalid = torch.zeros((G, 2), dtype=torch.int32, device=dev)
        n.cand, n.counts = field_ptr("cand"), field_ptr("counts")
        addressing(n)
        n.G, n.num_levels, n.topk = G, L, topk
        for l in range(L + 1):
            n.slot_off[l] = self.slot_off[l]
        n.do_nms, n.use_score3d = int(bool(inf.DO_NMS)), int(self.b3d_maps is not None)
        n.nms_thresh, n.post_topk = float(inf2.NMS_THRESH), int(inf2.POST_NMS_TOPK)
        # BEV stages (core.py:135-150, nuscenes_dd3d.py:423-465) run after the 2D NMS; the resize / clip / non-empty filter
        # of detector_postprocess sits between them, so it moves into whichever kernel comes at that point.
        n.do_postprocess = int(bool(inf.DO_POSTPROCESS) and not bev_single)
        n.out_size = field_ptr("outsize")
        n.sort_idx, n.sbox, n.scls = self.sort_idx.data_ptr(), self.sbox.data_ptr(), self.scls.data_ptr()
        n.mask, n.nvalid = self.mask.data_ptr(), self.nvalid.data_ptr()
        n.det, n.det_count, n.det_cap = self.det.data_ptr(), self.det_count.data_ptr(), self.det_cap
        self.nms_args = n
        self.nms_op = CallOp(lambda lib, st: hip.check(lib.dd3d_nms_finalize(C.byref(n), st), "nms_finalize"), "nms_finalize")
        self.ops.append(self.nms_op)
        if bev_single or bev_sample:
            # One BEV problem per call over the G images this rank finalises (the reference concatenates the batch: one
            # batched_nms_rotated, postprocessing.py:86-94).  Capacity: the LDS sorter holds 8192 BOXES -- actual detections, counted
            # on the device (<= POST_NMS_TOPK per image after the 2D stage: 81 images at 100); more trips the overflow flag, count_out
            # = -1, and collect() raises.
            ntot = G * self.det_cap
            ncapb = (ntot + 63) // 64 * 64
            self.bev_work = torch.zeros((ntot, 16), dtype=torch.float32, device=dev)
            self.bev_sbox = torch.zeros((ntot, 8), dtype=torch.float32, device=dev)
            self.bev_mask = torch.zeros((min(ncapb, 8192), ncapb // 64), dtype=torch.int64, device=dev)  # rows: sorted boxes (<= 8192)
            self.bev_meta = torch.zeros((4, ), dtype=torch.int32, device=dev)
            self.own_group = torch.arange(G, dtype=torch.int32, device=dev)  # dummy_group_idxs = {i: [i]} (core.py:137)
            if self.camera_sharded:  # sample membership is positional: cameras 6 s .. 6 s + 5 of the global order
                self.in_group = torch.arange(G, dtype=torch.int32, device=dev) // ncam
            self.bev_args = []
            self.det_stages = [(self.det, self.det_count)]  # every stage's buffers stay referenced: the arg structs hold raw pointers

            def stage(group, max_dets, write_global, do_pp, name):
                b = hip.BevArgs()
                det_out = torch.zeros_like(self.det)
                cnt_out = torch.zeros_like(self.det_count)
                b.det_in, b.count_in = self.det.data_ptr(), self.det_count.data_ptr()
                b.inv_K, b.pose, b.group = field_ptr("inv_K"), field_ptr("pose"), group.data_ptr()
                b.out_size = field_ptr("outsize")
                addressing(b)
                b.G, b.det_cap, b.num_classes = G, self.det_cap, C_
                b.iou_thresh, b.max_dets = float(inf.BEV_NMS_IOU_THRESH), int(max_dets)
                b.write_global, b.do_postprocess = int(write_global), int(do_pp)
                b.work, b.sbox, b.mask, b.meta = self.bev_work.data_ptr(), self.bev_sbox.data_ptr(), self.bev_mask.data_ptr(), self.bev_meta.data_ptr()
                b.det_out, b.count_out = det_out.data_ptr(), cnt_out.data_ptr()
                self.bev_args.append(b)
                self.ops.append(CallOp(lambda lib, st, b=b: hip.check(lib.dd3d_bev_nms_aggregate(C.byref(b), st), name), name))
                self.det, self.det_count = det_out, cnt_out  # what collect() reads
                self.det_stages.append((det_out, cnt_out))

            if bev_single:
                stage(self.own_group, 0, False, bool(inf.DO_POSTPROCESS), "bev_nms")
            if bev_sample:
                stage(self.in_group, int(model.max_num_dets_per_sample), True, False, "nusc_sample_aggregate")
                self.has_global_boxes = True

    def check_status(self):
        """With the exchange, the verdict is the OR over all ranks' records (delivered by the step's all_gather), so that every rank raises on
        the same step; the local words are cleared as well."""
        # (DenseDepthPlan shares this class without the post-processing half: no exchange, no gathered buffer)
        if not (getattr(self, "exchange", False) and self.math == hip.MATH_F16X2 and getattr(self, "gathered", None) is not None):
            return super().check_status()
        fl = self.gathered.view(self.world_size, self.record_len)[:, self.flags_off:self.flags_off + 2].view(torch.int32).cpu()
        over = [r for r in range(self.world_size) if int(fl[r, 0]) & hip.STATUS_F16_OVERFLOW]
        under = [r for r in range(self.world_size) if int(fl[r, 1])]
        if over or under:
            self.status.zero_()
            what = (f"an activation left the half range while being split (|x| > {65504.0 / self.act_scale:g} at plane scale {self.act_scale:g}) on "
                    f"rank(s) {over}" if over else
                    f"convolution outputs sit below the half range's useful part on rank(s) {under} (absolute floor {2.0**-25 / self.act_scale:.2g})")
            raise FloatingPointError(f"{what}: run this model with math='bf16x3' (every rank sees this verdict on the same step)")

    def gather_pairs(self):
        """(local record, gathered buffer [W x record]): the ONE tensor pair the multi-GPU step all-gathers between select/decode and
        the NMS stages."""
        return [(self.record, self.gathered)]

    def gathered_field(self, name):
        """Field `name` (record_fields) of every rank's images as delivered by the exchange: [W * B, words per image], rank-major =
        global image order.  counts come back as int32."""
        off, per = self.record_fields[name]
        g = self.gathered.view(self.world_size, self.record_len)[:, off:off + self.B * per]
        if name == "counts":
            g = g.view(torch.int32)
        return g.reshape(self.world_size * self.B, per)

    def gathered_counts(self):
        """Candidate counts [W*B, L] of every rank's images as delivered by the exchange (diagnostics / tests)."""
        return self.gathered_field("counts")

    def image_offset(self, g, name):
        """Word offset, relative to record 0's block of field `name`, of image g of the post stages -- the arithmetic of
        csrc/postproc.hip::rec_off (tests check the two against each other)."""
        per = self.record_fields[name][1]
        if not self.exchange:
            return g * per
        gg = self.img_first + g
        return (gg // self.B) * self.record_len + (gg % self.B) * per


class DenseDepthPlan(ForwardPlan):
    """Launch plan of DD3DDenseDepth (dense_depth.py:121-151): trunk, the box3d tower (one multi-segment launch per layer), the
    per-level 1-channel predictors with Scale / Offset folded in (one launch), then per level the aligned bilinear upsampling to
    the input resolution fused with the focal-length scaling."""
    def __init__(self, model, B, Hp, Wp, device=None, dry_run=False):
        PlanBase.__init__(self, device or model.device, dry_run=dry_run)
        self.adopt_weight_store(model)
        self._trunk(model, B, Hp, Wp)
        dev, feats, head = self.device, self.features, model.fcos3d_head
        L, Cf = len(feats), feats[0].C
        ping = [self.buf(f"ddA.{l}", f.B, f.H, f.W, Cf, kind="planes") for l, f in enumerate(feats)]
        pong = [self.buf(f"ddB.{l}", f.B, f.H, f.W, Cf, kind="planes") for l, f in enumerate(feats)]
        cur = list(feats)
        for i, conv in enumerate(head.box3d_tower):
            dst = ping if i % 2 == 0 else pong
            w, meta = self.pack(conv.weight)
            segs = []
            for l in range(L):
                norm = conv.norm[l] if isinstance(conv.norm, torch.nn.ModuleList) else conv.norm
                scale, shift = fold_norm(conv, norm)
                segs.append({"in": cur[l], "out": dst[l].view(), "w": w, "scale": self._vec(scale), "bias": self._vec(shift)})
                cur[l] = dst[l].view()
            self.ops.append(ConvOp(self, meta, 1, 1, segs, relu=True, name=f"dd_tower.{i}"))
        # predictors: a different filter per level (dense_depth.py:63-67,93-97), (conv + b) * scale + offset
        segs, self.dd_raw = [], []
        meta = None
        for l, conv in enumerate(head.dense_depth):
            w, meta = self.pack(conv.weight)
            b = conv.bias.detach().float().cpu() if conv.bias is not None else torch.zeros(1)
            sc = head.scales_depth[l].scale.detach().float().cpu() if head.use_scale else torch.ones(1)
            off = head.offsets_depth[l].bias.detach().float().cpu() if head.use_scale else torch.zeros(1)
            out = self.buf(f"dd_raw.{l}", feats[l].B, feats[l].H, feats[l].W, 4)
            self.dd_raw.append(out)
            segs.append({"in": cur[l], "out": out.view(0, 4), "w": w, "scale": self._vec(sc), "bias": self._vec(b * sc + off), "n_limit": 1})
        self.ops.append(ConvOp(self, meta, 1, 1, segs, relu=False, name="dd_predictors"))
        # upsample + focal scaling (tensor2d.py:28-47, dense_depth.py:140-151)
        self.depth_maps = []
        half = int(model.feature_locations_offset == "half")
        for l, f in enumerate(feats):
            stride = self.strides[l]
            assert f.H * stride == Hp and f.W * stride == Wp, "pyramid level does not tile the padded input"
            o = torch.zeros((B, Hp, Wp), dtype=torch.float32, device=dev)
            self.depth_maps.append(o)
            factor = float(model.scale_depth_by_focal_lengths_factor) if model.scale_depth_by_focal_lengths else 0.0

            def _up(lib, st, src=self.dd_raw[l], o=o, stride=stride, factor=factor, f=f):
                hip.check(lib.dd3d_aligned_bilinear_scale(src.t.data_ptr(), o.data_ptr(), self.inv_K.data_ptr(), B, f.H, f.W, 4, stride, half,
                                                          factor, st), "aligned_bilinear")

            self.ops.append(CallOp(_up, f"dd_upsample.{l}", dict(kind="aligned_bilinear_scale", src=self.dd_raw[l], out=o, factor=stride,
                                                                   offset_half=half, focal_factor=factor)))

