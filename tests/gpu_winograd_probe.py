"""Go / no-go probe for Winograd F(2x2, 3x3) on ONE head-tower layer in the f16x2 split arithmetic (round-5 verdict item 5; development tool).

    python tests/gpu_winograd_probe.py numerics          (CPU: error of the transform-domain f16x2 arithmetic against a float64 convolution)
    python tests/gpu_winograd_probe.py time [images]     (MI355X: the multiply stage with the library's own GEMM kernel + the bytes of the transforms)

Shapes: a tower layer is 3 towers x 5 levels of 256 -> 256 channels, 3 x 3 (fcos2d.py:137-152, fcos3d.py:163-180); at four images per launch
M = 4 x 10 230 output pixels per tower, K = 9 x 256.  F(2x2, 3x3) computes a 2 x 2 output tile from a 4 x 4 input tile with 16 products per
(input channel, output channel) instead of 36: per tower 16 GEMMs of [M / 4 tiles] x [256] x [256] -- 2.25 x fewer matrix products -- plus an input
transform V = B^T d B (16 values per tile and channel out of 16 inputs shared with the neighbours: 4 x the activation bytes), a filter transform
(once) and an output transform Y = A^T M A.

What the probe measures:
  numerics  the same layer three ways against float64: direct f16x2 (the shipped arithmetic: x * S = hi + lo halves, 3 products, f32 accumulate),
            Winograd in f32, Winograd with V and U split into half pairs (what an MFMA implementation would multiply).
  time      (a) the 16 x 3 GEMMs as ONE multi-segment 1 x 1 launch of the library's split-plane kernel on V-shaped operands (M / 4 rows, planes in,
            planes out): the multiply stage of a TWO-KERNEL implementation at the efficiency the shipped kernel reaches on such short-K problems
            (K = 256 = 8 K-tiles per tile); (b) a device copy moving the input transform's bytes (activation planes in, 4 x as many out): the
            floor of the transform kernel; (c) the shipped tower launch.  An in-kernel transform is priced in DESIGN section 4 (round 6).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def split_half(x, scale):
    """x * scale = hi + lo, two IEEE halves by round-to-nearest (float64 in, float32 terms out)."""
    xs = (x * scale).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def mm3(ah, al, bh, bl):
    """hi.hi + hi.lo + lo.hi with f32 accumulation (a: [m, k], b: [k, n])."""
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def numerics():
    rng = np.random.default_rng(5)
    C, N, H, W = 256, 256, 24, 80  # the p4 level of one image
    x = np.maximum(rng.standard_normal((C, H, W)), 0.0) * 1.5  # post-ReLU tower activations: half zeros, O(1) magnitudes
    w = rng.standard_normal((N, C, 3, 3)) / np.sqrt(9 * C)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    # float64 direct
    cols = np.stack([xp[:, dh:dh + H, dw:dw + W] for dh in range(3) for dw in range(3)], 1).reshape(C * 9, H * W)
    ref = (w.reshape(N, C * 9) @ cols).reshape(N, H, W)
    scale_ref = np.abs(ref).max()
    # direct f16x2: activations x 16, filter rows scaled into [2^13, 2^14)
    wrow = w.reshape(N, -1)
    rs = 2.0 ** (13 - np.floor(np.log2(np.abs(wrow).max(1))))
    xh, xl = split_half(cols, 16.0)
    wh, wl = split_half(wrow, rs[:, None])
    direct = mm3(wh, wl, xh, xl).astype(np.float64) / (16.0 * rs[:, None])
    e_direct = np.abs(direct.reshape(N, H, W) - ref).max() / scale_ref
    # Winograd tiles
    th, tw = H // 2, W // 2
    d = np.stack([[xp[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(tw)] for i in range(th)], 0)  # [th][tw][C][4][4]
    d = d.reshape(th * tw, C, 4, 4)
    V = np.einsum("ab,tcbd,ed->tcae", BT, d, BT)  # B^T d B, exact in float64 (sums of <= 4 inputs)
    U = np.einsum("ab,ncbd,ed->ncae", G, w, G)
    Mx = np.einsum("tcae,ncae->tnae", V, U)
    Y64 = np.einsum("ab,tnbd,ed->tnae", AT, Mx, AT)

    def untile(Y):
        return Y.reshape(th, tw, N, 2, 2).transpose(2, 0, 3, 1, 4).reshape(N, H, W)

    e_w64 = np.abs(untile(Y64) - ref).max() / scale_ref
    # Winograd, transforms in f32, products in f32 (what an f32 cuDNN Winograd does)
    V32 = np.einsum("ab,tcbd,ed->tcae", BT.astype(np.float32), d.astype(np.float32), BT.astype(np.float32))
    U32 = U.astype(np.float32)
    M32 = np.einsum("tcae,ncae->tnae", V32, U32)
    e_w32 = np.abs(untile(np.einsum("ab,tnbd,ed->tnae", AT.astype(np.float32), M32, AT.astype(np.float32)).astype(np.float64)) - ref).max() / scale_ref
    # Winograd with half-pair operands: V from the f32 value (hi + lo of the stored planes), re-split at plane scale 4 (|V| <= 4 |x|);
    # U row-scaled per (n, component) into [2^13, 2^14) and split
    V32s = V32.astype(np.float64)
    Msplit = np.zeros((th * tw, N, 4, 4), dtype=np.float64)
    for a in range(4):
        for e in range(4):
            u = U[:, :, a, e]
            us = 2.0 ** (13 - np.floor(np.log2(np.maximum(np.abs(u).max(1), 1e-30))))
            uh, ul = split_half(u, us[:, None])
            vh, vl = split_half(V32s[:, :, a, e], 4.0)
            Msplit[:, :, a, e] = mm3(vh, vl, uh.T, ul.T).astype(np.float64) / (4.0 * us[None, :])
    Ysplit = np.einsum("ab,tnbd,ed->tnae", AT.astype(np.float32), Msplit.astype(np.float32), AT.astype(np.float32)).astype(np.float64)
    e_wsplit = np.abs(untile(Ysplit) - ref).max() / scale_ref
    print(f"one tower layer, {C} -> {N} channels on a {H} x {W} map; max |error| / max |float64 result| ({scale_ref:.3f}):")
    print(f"  direct, f16x2 split (shipped)                       {e_direct:.3e}")
    print(f"  Winograd F(2x2,3x3), float64 transforms + products  {e_w64:.3e}")
    print(f"  Winograd, f32 transforms + f32 products             {e_w32:.3e}")
    print(f"  Winograd, f32 transforms, V and U as half pairs     {e_wsplit:.3e}   ({e_wsplit / e_direct:.1f} x the shipped arithmetic)")
    print(f"  largest |V| / largest |x| = {np.abs(V).max() / np.abs(x).max():.2f} (the half range shrinks by this factor: plane scale 16 -> 4)")


def timing(images=4):
    import __graft_entry__ as g
    g.build()
    from dd3d_amd import hip
    from dd3d_amd.engine import ConvOp, PlanBase, pack_filter
    plan = PlanBase("cuda")
    plan.math = hip.MATH_F16X2
    M = images * 10230
    tiles = M // 4
    gen = torch.Generator().manual_seed(3)
    segs, keep = [], []
    for tower in range(3):
        for comp in range(16):
            w = torch.randn(256, 256, 1, 1, generator=gen) / 16.0
            wp, meta = pack_filter(w, plan.device)
            vin = plan.buf(f"V.{tower}.{comp}", 1, 1, tiles, 256, kind="both")
            vin.t.copy_(torch.relu(torch.randn(1, 1, tiles, 256, generator=gen)).to(plan.device) * 3.0)
            plan.split(vin.view(), name=f"V.{tower}.{comp}.split")
            out = plan.buf(f"M.{tower}.{comp}", 1, 1, tiles, 256, kind="planes")
            segs.append({"in": vin.view(), "out": out.view(), "w": wp, "scale": torch.ones(256, device=plan.device), "bias": torch.zeros(256, device=plan.device)})
    plan.launch()  # the splits
    torch.cuda.synchronize()
    st = hip.current_stream()

    def time_op(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    flops = 2.0 * 48 * tiles * 256 * 256
    print(f"{images} images per launch: {tiles} tiles per tower; multiply stage = 48 GEMMs [{tiles} x 256] x [256 x 256] = {flops / 1e9:.1f} GFLOP "
          f"(direct: {2.0 * 3 * M * 256 * 2304 / 1e9:.1f} GFLOP)")
    best = None
    for tile in (None, hip.TILE_256x256_W8, hip.TILE_256x128, hip.TILE_128x128, hip.TILE_128x256_T24, hip.TILE_256x128_T42):
        try:
            op = ConvOp(plan, meta, 1, 0, segs, relu=False, tile=tile, name=f"winograd.multiply.{tile}", math=hip.MATH_F16X2)
            us = time_op(lambda: op(plan.lib, st))
            print(f"  multiply stage, tile {op.info['tile_name']:12s} split-K {op.info['splitk']}: {us:8.1f} us = {flops / us / 1e6:7.1f} TFLOP/s f32-equivalent "
                  f"({flops / us / 1e6 / 833.3:.3f} of the f16x2 roofline), {op.info['blocks']} blocks")
            best = us if best is None else min(best, us)
        except Exception as e:
            print(f"  tile {tile}: {type(e).__name__}: {str(e)[:100]}")
    act = torch.empty(3 * M * 256 * 2, dtype=torch.int16, device="cuda")      # activation planes of the three towers
    vbuf = torch.empty(4 * act.numel(), dtype=torch.int16, device="cuda")     # V: 16 values per 2 x 2 tile and channel
    t_copy = time_op(lambda: vbuf.view(4, -1).copy_(act.view(1, -1).expand(4, -1)))
    print(f"  input-transform bytes ({act.numel() * 2 / 1e6:.0f} MB in, {vbuf.numel() * 2 / 1e6:.0f} MB out) as a device copy: {t_copy:8.1f} us "
          f"({(act.numel() + vbuf.numel()) * 2 / t_copy / 1e6:.2f} TB/s) -- the floor of a transform KERNEL")
    print(f"  two-kernel Winograd layer >= {best + t_copy:.0f} us (multiply {best:.0f} + transform floor {t_copy:.0f}; output transform in the GEMM epilogue not counted)")
    return best, t_copy


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    else:
        numerics()
