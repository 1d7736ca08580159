"""The split-plane data flow (csrc/conv_planes.hip, conv_common.h::conv_epilogue) through the C ABI vs a plain PyTorch fp32 reference:
f32 -> dd3d_split_planes -> conv reading planes by LDS-DMA -> f32 and / or plane output -> the next conv reading those planes.

Tolerances: the three-term mode (bf16x3) is held to the f32 kernels' tolerance (2e-5 of max |ref|); the reduced modes to what their
operand width gives (bf16x2: two bf16 terms ~ 2^-17 per operand; bf16: 2^-9), stated per mode below; the two-half-term mode (f16x2:
2 x 11 bits inside the half range) is held to the f32 tolerance as well."""
import pytest
import torch
import torch.nn.functional as F

from dd3d_amd import hip

pytestmark = pytest.mark.gpu

MODES = {"bf16x3": (hip.MATH_BF16X3, 2e-5), "f16x2": (hip.MATH_F16X2, 2e-5), "bf16x2": (hip.MATH_BF16X2, 1e-4), "bf16": (hip.MATH_BF16, 3e-2)}

CASES = [
    # name, B, H, W, Cin, Cout, k, stride, pad, relu, residual, tile, splitk
    ("tower_256x128", 1, 33, 41, 256, 256, 3, 1, 1, True, True, hip.TILE_256x128, 1),
    ("tower_256x128_sk3_s2", 1, 30, 44, 128, 192, 3, 2, 1, False, False, hip.TILE_256x128, 3),
    ("k32_256x128", 1, 24, 40, 32, 128, 1, 1, 0, False, False, hip.TILE_256x128, 1),  # a single K-tile
    ("tower_128x128", 2, 17, 23, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128, 1),
    ("sk4_128x128", 1, 12, 20, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128, 4),
    ("tower_128x64", 1, 24, 40, 256, 256, 3, 1, 1, True, False, hip.TILE_128x64, 1),
    ("s2_odd_128x64_sk2", 1, 13, 21, 256, 256, 3, 2, 1, False, False, hip.TILE_128x64, 2),
    ("sk3_64x128", 1, 9, 31, 128, 192, 3, 1, 1, False, True, hip.TILE_64x128, 3),
    ("w4_128x128", 2, 17, 23, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128_W4, 1),
    ("w4_64x64", 1, 24, 40, 256, 256, 3, 1, 1, True, False, hip.TILE_64x64_W4, 1),
    ("w4_64x64_sk2_k64", 1, 24, 40, 64, 64, 1, 1, 0, False, True, hip.TILE_64x64_W4, 2),
    ("w4_128x64_sk4", 1, 24, 40, 256, 128, 3, 1, 1, True, True, hip.TILE_128x64_W4, 4),
    ("root1x1_k448", 1, 24, 40, 448, 128, 1, 1, 0, True, False, None, None),
    ("pred_n55", 1, 24, 40, 256, 55, 3, 1, 1, False, False, None, None),
    ("pred_n5", 1, 24, 40, 256, 5, 3, 1, 1, False, False, None, None),
    ("tiny_3x5", 2, 3, 5, 256, 256, 3, 1, 1, True, False, None, None),  # P7-sized: every tap row / column meets the border
    ("model_choice", 1, 48, 160, 256, 256, 3, 1, 1, True, False, None, None),
]


def _plan(math):
    from dd3d_amd.engine import PlanBase
    plan = PlanBase("cuda")
    plan.math = math
    return plan


ROW_CASES = [  # 3x3 / stride 1: the row-shared kernel (csrc/conv_planes_row.hip); the same shapes also run on the per-tap kernel below
    ("row_256x128_wraps", 2, 33, 41, 64, 128, 3, 1, 1, True, True, hip.TILE_256x128, 1),      # 41-wide rows: every wave crosses image rows
    ("row_256x128_sk3", 1, 30, 44, 96, 192, 3, 1, 1, False, False, hip.TILE_256x128, 3),       # 27 K-tiles / 3 = 9: slices start on a filter row
    ("row_sk2_not_on_a_row", 1, 12, 20, 64, 64, 3, 1, 1, False, False, hip.TILE_64x64_W4, 4),  # 18 K-tiles / 4: falls back to the per-tap kernel
    ("row_64x64w4_b3", 3, 7, 9, 256, 256, 3, 1, 1, True, False, hip.TILE_64x64_W4, 1),         # three tiny images: batch boundaries inside a tile
    ("row_128x64w4_sk3", 1, 24, 40, 128, 128, 3, 1, 1, True, True, hip.TILE_128x64_W4, 3),
    ("row_64x128_1x3", 1, 1, 3, 32, 128, 3, 1, 1, False, False, hip.TILE_64x128, 1),           # a 1 x 3 image: every neighbour but two is padding
    # wave tiles of 128 x 64 / 64 x 128 outputs (round 3): 4-wave blocks with 8 accumulator blocks per wave, and the 8-wave 256 x 256 tile
    ("row_t42_wraps", 2, 33, 41, 64, 256, 3, 1, 1, True, True, hip.TILE_256x128_T42, 1),
    ("row_t42_sk3", 1, 30, 44, 96, 192, 3, 1, 1, False, False, hip.TILE_256x128_T42, 3),
    ("row_t24_b3", 3, 7, 9, 256, 256, 3, 1, 1, True, False, hip.TILE_128x256_T24, 1),
    ("row_t24_n320", 1, 24, 40, 64, 320, 3, 1, 1, False, True, hip.TILE_128x256_T24, 1),       # two N tiles, the second half empty
    ("row_w8_256x256", 2, 17, 23, 128, 256, 3, 1, 1, True, False, hip.TILE_256x256_W8, 1),  # (this tile carries no residual: registers)
    ("row_w8_192x256", 2, 17, 23, 128, 256, 3, 1, 1, True, False, hip.TILE_192x256_W8, 1),  # (round 6: three accumulator row blocks per wave; row kernel only)
    ("row_w8_192x256_n320", 1, 40, 44, 64, 320, 3, 1, 1, False, False, hip.TILE_192x256_W8, 1),  # ten m-tiles, two N tiles (the second a quarter full)
    ("t42_1x1_k448", 1, 24, 40, 448, 128, 1, 1, 0, True, False, hip.TILE_256x128_T42, 1),
    ("t24_s2_sk2", 1, 13, 21, 256, 256, 3, 2, 1, False, False, hip.TILE_128x256_T24, 2),
    ("w8_1x1", 1, 24, 40, 256, 512, 1, 1, 0, False, False, hip.TILE_256x256_W8, 1),
]


@pytest.mark.parametrize("row_kernel", [1, 0], ids=["rowshared", "pertap"])
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", CASES + ROW_CASES, ids=[c[0] for c in CASES + ROW_CASES])
def test_planes_conv_matches_torch(hiplib, case, mode, row_kernel, monkeypatch):
    from dd3d_amd.engine import ConvOp, pack_filter
    name, B, H, W, Cin, Cout, k, stride, pad, relu, use_res, tile, splitk = case
    if not row_kernel and not (k == 3 and stride == 1):
        pytest.skip("only 3x3 / stride 1 has two kernels")
    monkeypatch.setenv("DD3D_CONV_ROW", str(row_kernel))
    math, rtol = MODES[mode]
    if tile in (hip.TILE_256x256_W8, hip.TILE_192x256_W8) and hip.MATH_PLANES[math] > 2:
        pytest.skip("the 8-wave 256-column tiles exist for the one- and two-term modes only (register file)")
    if tile == hip.TILE_192x256_W8 and not row_kernel:
        pytest.skip("instantiated for the row-shared kernel only")
    g = torch.Generator().manual_seed(sum(map(ord, name)) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k)**0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    ref = F.conv2d(x, w, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)

    plan = _plan(math)
    wp, meta = pack_filter(w, plan.device)
    # the input is a 32-aligned channel slice of a wider buffer (concat by placement carries over to the chunk-major plane layout)
    xin = plan.buf("x", B, H, W, Cin + 64, kind="both")
    xin.t[..., 32:32 + Cin] = x.permute(0, 2, 3, 1).to(plan.device)
    plan.split(xin.view(32, Cin), name="x.split")
    cpad = (Cout + 31) // 32 * 32
    yout = plan.buf("y", B, Ho, Wo, cpad + 32, kind="both")
    yout.t.fill_(-777.0)
    yout.p.fill_(0x1234)
    seg = {"in": xin.view(32, Cin), "out": yout.view(32, cpad), "w": wp, "scale": scale.to(plan.device), "bias": bias.to(plan.device)}
    if res is not None:
        rbuf = plan.buf("r", B, Ho, Wo, Cout)
        rbuf.t.copy_(res.permute(0, 2, 3, 1))
        seg["res"] = rbuf.view()
    op = ConvOp(plan, meta, stride, pad, [seg], relu, tile=tile, splitk=splitk, name=name, math=math)
    assert op.math == math and op.in_planes
    plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    got = yout.t[..., 32:32 + Cout].permute(0, 3, 1, 2).cpu()
    tol = rtol * max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= tol, f"{name}/{mode}: max abs err {err:.3e} > {tol:.3e} (info {op.info})"
    # f32 channels outside [32, 32 + Cout) untouched
    assert torch.all(yout.t[..., :32] == -777.0) and torch.all(yout.t[..., 32 + Cout:] == -777.0)
    # plane output: chunk images of the slice hold the split of the f32 output (exactly, for the three-term split; channels past N in
    # the last chunk are zero); the chunk before the slice is untouched
    np_ = hip.MATH_PLANES[math]
    assert torch.all(yout.p[0] == 0x1234)
    if math == hip.MATH_F16X2:
        terms = yout.p[1:].view(torch.float16).float() / yout.plane_scale
    else:
        terms = (yout.p[1:].to(torch.int32) << 16).view(torch.float32)
    dec = terms.sum(2).permute(1, 0, 2).reshape(B, Ho, Wo, cpad).permute(0, 3, 1, 2).cpu()
    assert torch.all(dec[:, Cout:] == 0)
    if math == hip.MATH_BF16X3:
        assert torch.equal(dec[:, :Cout], got)
    elif math == hip.MATH_F16X2:  # 22 bits, down to the absolute floor 2^-25 / plane scale
        assert float((dec[:, :Cout] - got).abs().max()) <= 2.0**-21 * float(got.abs().max()) + 2.0**-24 / yout.plane_scale
        assert int(plan.status.cpu()) == 0
    else:
        step = {2: 2.0**-15, 1: 2.0**-8}[np_]
        assert float(((dec[:, :Cout] - got).abs() / got.abs().clamp(min=1e-20)).max()) <= step


@pytest.mark.parametrize("mode", list(MODES))
def test_two_convs_chained_through_planes_only(hiplib, mode):
    """conv -> (planes only, no f32 copy) -> conv, multi-segment like the head towers (per-segment scale / bias), against torch."""
    from dd3d_amd.engine import ConvOp, pack_filter
    math, rtol = MODES[mode]
    plan = _plan(math)
    g = torch.Generator().manual_seed(11)
    w1 = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    w2 = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    wp1, meta = pack_filter(w1, plan.device)
    wp2, _ = pack_filter(w2, plan.device)
    shapes = [(12, 40), (6, 20), (3, 10), (2, 5), (1, 3)]
    segs1, segs2, refs, outs = [], [], [], []
    for l, (h, wd) in enumerate(shapes):
        x = torch.randn(2, 256, h, wd, generator=g)
        s1, b1 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
        s2, b2 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
        y1 = F.relu(F.conv2d(x, w1, None, padding=1) * s1.view(1, -1, 1, 1) + b1.view(1, -1, 1, 1))
        refs.append(F.relu(F.conv2d(y1, w2, None, padding=1) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1)))
        xb = plan.buf(f"x{l}", 2, h, wd, 256, kind="both")
        xb.t.copy_(x.permute(0, 2, 3, 1))
        plan.split(xb.view(), name=f"x{l}.split")
        mid = plan.buf(f"m{l}", 2, h, wd, 256, kind="planes")
        yb = plan.buf(f"y{l}", 2, h, wd, 256, kind="both")
        dev = plan.device
        segs1.append({"in": xb.view(), "out": mid.view(), "w": wp1, "scale": s1.to(dev), "bias": b1.to(dev)})
        segs2.append({"in": mid.view(), "out": yb.view(), "w": wp2, "scale": s2.to(dev), "bias": b2.to(dev)})
        outs.append(yb)
    for segs, nm in ((segs1, "l1"), (segs2, "l2")):
        op = ConvOp(plan, meta, 1, 1, segs, relu=True, name=nm, math=math)
        assert op.in_planes
        plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    for l, ref in enumerate(refs):
        got = outs[l].t.permute(0, 3, 1, 2).cpu()
        err = (got - ref).abs().max().item()
        assert err <= 2 * rtol * max(1.0, ref.abs().max().item()), (l, mode, err)


def test_split_planes_relu_and_slices(hiplib):
    """dd3d_split_planes: exact three-term split, RNE terms of the reduced modes, the rectified variant (LastLevelP6P7) and a channel
    slice source with a pitch."""
    for mode, (math, _) in MODES.items():
        plan = _plan(math)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 5, 7, 96, generator=g) * torch.logspace(-6, 6, 96).view(1, 1, 1, 96)
        src = plan.buf("s", 2, 5, 7, 160, kind="f32")
        src.t[..., 32:128] = x.to(plan.device)
        dst = plan.buf("d", 2, 5, 7, 96, kind="planes")
        dstr = plan.buf("dr", 2, 5, 7, 96, kind="planes")
        plan.split(src.view(32, 96), dst=dst.view(), name="s")
        plan.split(src.view(32, 96), relu=True, dst=dstr.view(), name="sr")
        plan.launch()
        torch.cuda.synchronize()
        got, gotr = dst.nchw().permute(0, 2, 3, 1).cpu(), dstr.nchw().permute(0, 2, 3, 1).cpu()
        if math == hip.MATH_BF16X3:
            assert torch.equal(got, x) and torch.equal(gotr, F.relu(x))
        elif math == hip.MATH_F16X2:  # magnitudes 1e-6 .. 1e6 at plane scale 16: the top of the range overflows and is flagged
            ok = x.abs() * plan.act_scale <= 65504.0
            err = ((got - x).abs() - 2.0**-24 / plan.act_scale).clamp(min=0) / x.abs()
            assert float(err[ok].max()) <= 2.0**-22 and int(plan.status.cpu()) == hip.STATUS_F16_OVERFLOW
            with pytest.raises(FloatingPointError, match="half range"):
                plan.check_status()
            assert int(plan.status.cpu()) == 0
        else:
            hi = x.to(torch.bfloat16)
            want = hi.float() + ((x - hi.float()).to(torch.bfloat16).float() if math == hip.MATH_BF16X2 else 0.0)
            assert torch.equal(got, want), mode
            assert torch.equal(gotr, torch.where(x > 0, want, torch.zeros_like(want))), mode


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "bf16x2"])
def test_pool_and_topdown_write_their_planes(hiplib, mode):
    """dd3d_maxpool2x2_planes / dd3d_upsample2x_add_planes: the f32 result equals the plain kernels' (bit-exact vs torch), and the
    planes written in the same launch decode to it (exactly for the three-term split)."""
    math, _ = MODES[mode]
    plan = _plan(math)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 12, 20, generator=g)
    xb = plan.buf("x", 2, 12, 20, 96, kind="f32")
    xb.t[..., 32:96] = x.permute(0, 2, 3, 1).to(plan.device)
    yb = plan.buf("y", 2, 6, 10, 128, kind="both")  # pooled map placed at channels 32..95 of a wider (concat) buffer
    plan.maxpool(xb.view(32, 64), yb.view(32, 64))
    c = torch.randn(2, 64, 3, 5, generator=g)
    cb = plan.buf("c", 2, 3, 5, 64, kind="f32")
    cb.t.copy_(c.permute(0, 2, 3, 1))
    plan.upsample_add(yb.view(32, 64), cb.view())
    assert len(plan.ops) == 2  # no separate split launches
    plan.launch()
    torch.cuda.synchronize()
    ref = F.max_pool2d(x, 2, 2) + F.interpolate(c, scale_factor=2.0, mode="nearest")
    got = yb.t[..., 32:96].permute(0, 3, 1, 2).cpu()
    assert torch.equal(got, ref)
    if math == hip.MATH_F16X2:
        terms = yb.p[1:3].view(torch.float16).float() / yb.plane_scale
    else:
        terms = (yb.p[1:3].to(torch.int32) << 16).view(torch.float32)
    dec = terms.sum(2).permute(1, 0, 2).reshape(2, 6, 10, 64).permute(0, 3, 1, 2).cpu()
    if math == hip.MATH_BF16X3:
        assert torch.equal(dec, ref)
    else:
        tol = {hip.MATH_F16X2: 2.0**-21, hip.MATH_BF16X2: 2.0**-15}[math]
        assert float((dec - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-8
    assert torch.all(yb.p[0] == 0) and torch.all(yb.p[3] == 0)  # neighbouring chunk images untouched


RES_CASES = [
    # name, B, H, W, Cin, Cout, k, relu, tile, splitk
    ("block2_128x64w4", 2, 24, 40, 64, 64, 3, True, hip.TILE_128x64_W4, 1),       # DLA level 2 BasicBlock conv2
    ("block3_64x64w4", 1, 12, 20, 128, 128, 3, True, hip.TILE_64x64_W4, 1),
    ("block5_sk3_128x64", 1, 6, 10, 96, 160, 3, True, hip.TILE_128x64, 3),         # split-K + residual, a partly filled last chunk
    ("lateral_1x1_256x128", 2, 12, 20, 128, 256, 1, False, hip.TILE_256x128, 1),   # FPN lateral
    ("lateral_1x1_t42", 1, 24, 40, 64, 256, 1, False, hip.TILE_256x128_T42, 1),
    ("lateral_model_choice", 1, 48, 160, 128, 256, 1, False, None, None),
]


@pytest.mark.parametrize("res_form", ["f32", "planes", "planes_up"])
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", RES_CASES, ids=[c[0] for c in RES_CASES])
def test_residual_forms(hiplib, case, mode, res_form):
    """dd3d_conv_seg.res_mode 1 / 2 / 3 (ABI 4): the residual as an f32 map, as split planes of the same pixel (dla.py:59-60 without an
    f32 twin of the source) and as the split planes of the map at half the resolution (FPN top-down sum [ext] in the lateral's epilogue);
    the output is written as planes ONLY (what the planes-only data flow does) and decoded for the comparison."""
    from dd3d_amd.engine import ConvOp, pack_filter
    name, B, H, W, Cin, Cout, k, relu, tile, splitk = case
    math, rtol = MODES[mode]
    g = torch.Generator().manual_seed(sum(map(ord, name)) % 1000 + len(res_form))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k)**0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    up = res_form == "planes_up"
    res = torch.randn(B, Cout, H // 2 if up else H, W // 2 if up else W, generator=g)
    plan = _plan(math)
    wp, meta = pack_filter(w, plan.device)
    xin = plan.buf("x", B, H, W, Cin, kind="both")
    xin.t.copy_(x.permute(0, 2, 3, 1))
    plan.split(xin.view(), name="x.split")
    cpad = (Cout + 31) // 32 * 32
    # the residual source is a 32-aligned slice of a wider buffer
    rbuf = plan.buf("r", B, res.shape[2], res.shape[3], cpad + 32, kind="f32" if res_form == "f32" else "both")
    rbuf.t[..., 32:32 + Cout] = res.permute(0, 2, 3, 1).to(plan.device)
    if res_form != "f32":
        plan.split(rbuf.view(32, cpad), name="r.split")
    yout = plan.buf("y", B, H, W, cpad, kind="planes")
    seg = {"in": xin.view(), "out": yout.view(), "w": wp, "scale": scale.to(plan.device), "bias": bias.to(plan.device), "res": rbuf.view(32, cpad),
           "res_up": up}
    op = ConvOp(plan, meta, 1, k // 2, [seg], relu, tile=tile, splitk=splitk, name=name, math=math)
    assert op.in_planes and op.res_forms == [res_form]
    plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    assert int(plan.status.cpu()) == 0
    r = rbuf.nchw(32, Cout).cpu() if res_form == "f32" else None
    if r is None:  # what the planes hold (lossy in the reduced modes)
        if math == hip.MATH_F16X2:
            terms = rbuf.p[1:].view(torch.float16).float() / rbuf.plane_scale
        else:
            terms = (rbuf.p[1:].to(torch.int32) << 16).view(torch.float32)
        r = terms.sum(2).permute(1, 0, 2).reshape(B, res.shape[2], res.shape[3], cpad).permute(0, 3, 1, 2)[:, :Cout].cpu()
    if up:
        r = F.interpolate(r, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(x, w, None, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + r
    if relu:
        ref = F.relu(ref)
    got = yout.nchw().cpu()
    assert torch.all(got[:, Cout:] == 0)
    # the output exists as planes only: its own rounding on top of the convolution's
    out_step = {hip.MATH_BF16X3: 0.0, hip.MATH_F16X2: 2.0**-21, hip.MATH_BF16X2: 2.0**-15, hip.MATH_BF16: 2.0**-8}[math]
    tol = (rtol + out_step) * max(1.0, ref.abs().max().item())
    err = (got[:, :Cout] - ref).abs().max().item()
    assert err <= tol, f"{name}/{mode}/{res_form}: max abs err {err:.3e} > {tol:.3e} (info {op.info})"


@pytest.mark.parametrize("mode", list(MODES))
def test_pool_of_planes_copies_the_winners_terms(hiplib, mode):
    """dd3d_maxpool2x2_planes_in (ABI 4): the pooled planes decode to max_pool2d of the decoded input planes, bit for bit, for a channel
    slice placed in a wider (concat) buffer."""
    math, _ = MODES[mode]
    plan = _plan(math)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 96, 10, 14, generator=g)
    x[:, :, :2] = -x[:, :, :2].abs()  # windows of negative values only
    xb = plan.buf("x", 3, 10, 14, 160, kind="both")
    xb.t[..., 32:128] = x.permute(0, 2, 3, 1).to(plan.device)
    plan.split(xb.view(32, 96), name="x.split")
    xin = plan.buf("xp", 3, 10, 14, 96, kind="planes")
    yb = plan.buf("y", 3, 5, 7, 160, kind="planes")
    plan.launch()
    torch.cuda.synchronize()
    xin.p.copy_(xb.p[1:4])
    yb.p.fill_(0x0777)
    plan.ops[:] = []
    plan.maxpool(xin.view(), yb.view(64, 96))
    assert len(plan.ops) == 1
    plan.launch()
    torch.cuda.synchronize()
    want = F.max_pool2d(xin.nchw().cpu(), 2, 2)
    got = yb.nchw(64, 96).cpu()
    assert torch.equal(got, want)
    assert torch.all(yb.p[:2] == 0x0777)  # chunk images before the slice untouched


def test_split_modes_against_float64(hiplib):
    """Error of every arithmetic mode against a float64 convolution of the same f32 data.  The two f32-equivalent split modes -- bf16x3
    (three bf16 terms, 6 products) and f16x2 (two IEEE-half terms, 3 products) -- must sit at f32 rounding level like the exact-f32 MFMA
    kernel (measured on MI355X: f32 4.1e-7, bf16x3 3.7e-7, f16x2 3.4e-7 of max |ref|; bf16x2 4.7e-6; bf16 2.2e-3)."""
    from dd3d_amd.engine import ConvOp, pack_filter
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 1, 24, 40, 256, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 48.0
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    err = {}
    for name, math in [("f32", hip.MATH_F32)] + [(k, v[0]) for k, v in MODES.items()]:
        plan = _plan(math)
        wp, meta = pack_filter(w, plan.device)
        xin, yout = plan.buf("x", B, H, W, Cin, kind="both"), plan.buf("y", B, H, W, Cout)
        xin.t.copy_(x.permute(0, 2, 3, 1))
        if plan.use_planes:
            plan.split(xin.view(), name="x.split")
        ones, zeros = torch.ones(Cout, device=plan.device), torch.zeros(Cout, device=plan.device)
        plan.ops.append(ConvOp(plan, meta, 1, 1, [{"in": xin.view(), "out": yout.view(), "w": wp, "scale": ones, "bias": zeros}], False, name="acc"))
        plan.launch()
        torch.cuda.synchronize()
        err[name] = float((yout.nchw().cpu().double() - ref).abs().max() / ref.abs().max())
    print("max |err| / max |ref| vs float64: " + "  ".join(f"{k} {v:.2e}" for k, v in err.items()))
    assert err["f32"] < 2e-6
    for mode in ("bf16x3", "f16x2"):
        assert err[mode] < 2e-6 and err[mode] < 3 * err["f32"] + 2e-7, (mode, err)
    assert err["bf16x2"] < 5e-5 and 1e-4 < err["bf16"] < 2e-2


def test_f16x2_keeps_small_and_large_magnitudes(hiplib):
    """The half-term split inside its range: activations spanning 1e-4 .. 3.5e3 (plane scale 16: the half format holds 65504 / 16) and
    filters spanning six decades per row stay at f32-level accuracy relative to the result, because the per-row filter scale and the
    plane scale are powers of two that leave through the epilogue exactly."""
    from dd3d_amd.engine import ConvOp, pack_filter
    g = torch.Generator().manual_seed(9)
    B, H, W, Cin, Cout = 1, 12, 20, 128, 64
    x = torch.randn(B, Cin, H, W, generator=g) * torch.logspace(-4, 2.9, Cin).view(1, Cin, 1, 1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * torch.logspace(-6, 0, Cout).view(Cout, 1, 1, 1)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    plan = _plan(hip.MATH_F16X2)
    wp, meta = pack_filter(w, plan.device)
    xin, yout = plan.buf("x", B, H, W, Cin, kind="both"), plan.buf("y", B, H, W, Cout)
    xin.t.copy_(x.permute(0, 2, 3, 1))
    plan.split(xin.view(), name="x.split")
    ones, zeros = torch.ones(Cout, device=plan.device), torch.zeros(Cout, device=plan.device)
    plan.ops.append(ConvOp(plan, meta, 1, 1, [{"in": xin.view(), "out": yout.view(), "w": wp, "scale": ones, "bias": zeros}], False, name="range"))
    plan.launch()
    torch.cuda.synchronize()
    assert int(plan.status.cpu()) == 0
    got = yout.nchw().cpu().double()
    # per output channel (rows differ by six decades): error relative to that channel's largest output
    rel = ((got - ref).abs().amax((0, 2, 3)) / ref.abs().amax((0, 2, 3)))
    assert float(rel.max()) < 3e-6, rel
