"""HIP implicit-GEMM conv, f32 NHWC input kernels (through the C ABI) vs a plain PyTorch fp32 reference (F.conv2d on CPU)."""
import pytest
import torch
import torch.nn.functional as F

from dd3d_amd import hip

pytestmark = pytest.mark.gpu

CASES = [
    # name, B, H, W, Cin, Cout, k, stride, pad, relu, residual, tile, splitk
    ("tower3x3", 1, 24, 40, 256, 256, 3, 1, 1, True, False, None, None),
    ("tower3x3_128x128", 2, 17, 23, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128, 1),
    ("tower3x3_64x64_sk4", 1, 12, 20, 256, 256, 3, 1, 1, True, True, hip.TILE_64x64, 4),
    ("tower3x3_128x64_sk3", 1, 9, 31, 128, 128, 3, 1, 1, False, False, hip.TILE_128x64, 3),
    ("tower3x3_64x128", 1, 9, 31, 64, 192, 3, 1, 1, False, False, hip.TILE_64x128, 1),
    ("stride2", 1, 32, 48, 64, 128, 3, 2, 1, True, False, None, None),
    ("stride2_odd", 1, 13, 21, 256, 256, 3, 2, 1, False, False, None, None),
    ("root1x1", 1, 24, 40, 448, 128, 1, 1, 0, True, False, None, None),
    ("proj1x1", 1, 24, 40, 32, 64, 1, 1, 0, False, False, None, None),
    ("base7x7_c3", 1, 40, 72, 3, 16, 7, 1, 3, True, False, None, None),
    ("level0_c16", 1, 40, 72, 16, 16, 3, 1, 1, True, False, None, None),
    ("level1_c16_s2", 1, 40, 72, 16, 32, 3, 2, 1, True, False, None, None),
    ("pred_n5", 1, 24, 40, 256, 5, 3, 1, 1, False, False, None, None),
    ("pred_n55", 1, 24, 40, 256, 55, 3, 1, 1, False, False, None, None),
]


X3_CASES = [  # split-bf16 arithmetic (DD3D_MATH_BF16X3): same f32-level tolerance as the f32-MFMA kernel
    ("x3_tower3x3_128x128", 2, 17, 23, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128, 1),
    ("x3_tower3x3_128x64", 1, 24, 40, 256, 256, 3, 1, 1, True, False, hip.TILE_128x64, 1),
    ("x3_tower3x3_64x128_sk3", 1, 9, 31, 128, 192, 3, 1, 1, False, True, hip.TILE_64x128, 3),
    ("x3_128x128_sk4", 1, 12, 20, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128, 4),
    ("x3_stride2_odd", 1, 13, 21, 256, 256, 3, 2, 1, False, False, hip.TILE_128x64, 2),
    ("x3_root1x1", 1, 24, 40, 448, 128, 1, 1, 0, True, False, None, None),
    ("x3_proj1x1_k32", 1, 24, 40, 32, 64, 1, 1, 0, False, False, None, None),
    ("x3_pred_n55", 1, 24, 40, 256, 55, 3, 1, 1, False, False, None, None),
    ("x3_model_choice", 1, 48, 160, 256, 256, 3, 1, 1, True, False, None, None),
    ("x3_256x128", 1, 33, 41, 256, 256, 3, 1, 1, True, True, hip.TILE_256x128, 1),
    ("x3_256x128_sk3_s2", 1, 30, 44, 128, 192, 3, 2, 1, False, False, hip.TILE_256x128, 3),
    ("x3_256x128_k32", 1, 24, 40, 32, 128, 1, 1, 0, False, False, hip.TILE_256x128, 1),
    ("x3_128x128w4", 2, 17, 23, 256, 256, 3, 1, 1, True, True, hip.TILE_128x128_W4, 1),
    ("x3_128x128w4_sk3_s2", 1, 30, 44, 128, 192, 3, 2, 1, False, False, hip.TILE_128x128_W4, 3),
    ("x3_64x64w4", 1, 24, 40, 256, 256, 3, 1, 1, True, False, hip.TILE_64x64_W4, 1),
    ("x3_64x64w4_sk2_k32", 1, 24, 40, 64, 64, 1, 1, 0, False, True, hip.TILE_64x64_W4, 2),
    ("x3_128x64w4", 1, 13, 21, 256, 256, 3, 2, 1, False, False, hip.TILE_128x64_W4, 1),
    ("x3_128x64w4_sk4", 1, 24, 40, 256, 128, 3, 1, 1, True, True, hip.TILE_128x64_W4, 4),
    ("x3_128x64k2", 1, 24, 40, 256, 128, 3, 1, 1, True, True, hip.TILE_128x64_K2, 1),
    ("x3_128x64k2_odd_tiles_sk3", 1, 13, 21, 32, 128, 3, 2, 1, False, False, hip.TILE_128x64_K2, 3),  # 9 K-tiles: odd tails
    ("x3_64x128k2_k32", 1, 24, 40, 32, 128, 1, 1, 0, False, True, hip.TILE_64x128_K2, 1),  # a single K-tile: second half of the step is zero
    ("x3_64x64w4k2_sk2", 1, 24, 40, 448, 128, 1, 1, 0, True, False, hip.TILE_64x64_W4K2, 2),
    ("x3_64x64w4k2", 2, 17, 23, 128, 192, 3, 1, 1, False, True, hip.TILE_64x64_W4K2, 1),
]


@pytest.mark.parametrize("case", CASES + X3_CASES, ids=[c[0] for c in CASES + X3_CASES])
def test_conv_matches_torch(hiplib, case):
    from dd3d_amd.engine import ConvOp, PlanBase, pack_filter
    name, B, H, W, Cin, Cout, k, stride, pad, relu, use_res, tile, splitk = case
    math = hip.MATH_BF16X3 if name.startswith("x3_") else hip.MATH_F32
    g = torch.Generator().manual_seed(hash(name) % 1000)
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

    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    wp, meta = pack_filter(w, plan.device)
    # input lives in a wider buffer at a channel offset to exercise the pitch addressing
    cin_p = meta["Cin"]
    xin = plan.buf("x", B, H, W, cin_p + 8)
    xin.t[..., 4:4 + Cin] = x.permute(0, 2, 3, 1).to(plan.device)
    out_pitch = (Cout + 3) // 4 * 4 + 4
    yout = plan.buf("y", B, Ho, Wo, out_pitch)
    yout.t.fill_(-777.0)
    seg = {"in": xin.view(4, cin_p), "out": yout.view(4, out_pitch - 4), "w": wp, "scale": scale.to(plan.device), "bias": bias.to(plan.device)}
    if res is not None:
        rbuf = plan.buf("r", B, Ho, Wo, Cout)
        rbuf.t.copy_(res.permute(0, 2, 3, 1))
        seg["res"] = rbuf.view()
    op = ConvOp(plan, meta, stride, pad, [seg], relu, tile=tile, splitk=splitk, name=name, math=math)
    assert op.math == math
    plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    got = yout.t[..., 4:4 + Cout].permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs().max().item()
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    assert err <= tol, f"{name}: max abs err {err:.3e} > {tol:.3e} (info {op.info})"
    # channels outside the output slice must be untouched
    assert torch.all(yout.t[..., :4] == -777.0) and torch.all(yout.t[..., 4 + Cout:] == -777.0)


def test_multi_segment_launch(hiplib):
    """Several levels with their own (scale, bias) in ONE launch == the per-level BatchNorm of the shared towers."""
    from dd3d_amd.engine import ConvOp, PlanBase, pack_filter
    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    g = torch.Generator().manual_seed(7)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    wp, meta = pack_filter(w, plan.device)
    shapes = [(12, 40), (6, 20), (3, 10), (2, 5), (1, 3)]
    segs, refs, outs = [], [], []
    for l, (h, wd) in enumerate(shapes):
        x = torch.randn(2, 256, h, wd, generator=g)
        sc, bi = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
        xb, yb = plan.buf(f"x{l}", 2, h, wd, 256), plan.buf(f"y{l}", 2, h, wd, 256)
        xb.t.copy_(x.permute(0, 2, 3, 1))
        segs.append({"in": xb.view(), "out": yb.view(), "w": wp, "scale": sc.to(plan.device), "bias": bi.to(plan.device)})
        refs.append(F.relu(F.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)))
        outs.append(yb)
    plan.ops.append(ConvOp(plan, meta, 1, 1, segs, True, name="multi"))
    plan.launch()
    torch.cuda.synchronize()
    for yb, ref in zip(outs, refs):
        assert (yb.nchw().cpu() - ref).abs().max().item() < 5e-5


def test_aux_kernels(hiplib):
    from dd3d_amd.engine import PlanBase
    import ctypes as C
    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 12, 20, generator=g)
    xb, yb = plan.buf("x", 2, 12, 20, 40), plan.buf("y", 2, 6, 10, 36)
    xb.t[..., 8:40] = x.permute(0, 2, 3, 1).to(plan.device)
    plan.maxpool(xb.view(8, 32), yb.view(4, 32))
    c = torch.randn(2, 32, 3, 5, generator=g)
    cb = plan.buf("c", 2, 3, 5, 32)
    cb.t.copy_(c.permute(0, 2, 3, 1))
    plan.upsample_add(yb.view(4, 32), cb.view())
    plan.launch()
    torch.cuda.synchronize()
    ref = F.max_pool2d(x, 2, 2) + F.interpolate(c, scale_factor=2.0, mode="nearest")
    assert torch.equal(yb.nchw(4, 32).cpu(), ref)
    # preprocess + intrinsics inverse
    img = torch.randint(0, 256, (2, 3, 16, 24), dtype=torch.uint8, generator=g)
    sizes = torch.tensor([[16, 24], [11, 19]], dtype=torch.int32)
    mean, std = [103.53, 116.28, 123.675], [57.375, 57.12, 58.395]
    dst = torch.empty((2, 16, 24, 4), device="cuda")
    img_d, sizes_d = img.cuda(), sizes.cuda()  # keep the device copies alive until the kernels have run
    hip.check(hiplib.dd3d_preprocess_u8_nhwc4(img_d.data_ptr(), sizes_d.data_ptr(), dst.data_ptr(), 2, 16, 24,
                                              (C.c_float * 3)(*mean), (C.c_float * 3)(*std), hip.current_stream()))
    K = torch.tensor([[[748.7, 0.3, 632.5], [0, 748.8, 179.4], [0, 0, 1.0]], [[1260.9, 0, 812.7], [0, 1260.8, 489.3], [0, 0, 1]]])
    invK = torch.empty((2, 9), device="cuda")
    K_d = K.cuda()
    hip.check(hiplib.dd3d_invert_intrinsics(K_d.data_ptr(), invK.data_ptr(), 2, hip.current_stream()))
    torch.cuda.synchronize()
    ref = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    ref[1, :, 11:, :] = 0
    ref[1, :, :, 19:] = 0
    assert torch.equal(dst[..., :3].permute(0, 3, 1, 2).cpu(), ref) and torch.all(dst[..., 3] == 0)
    assert (invK.view(2, 3, 3).cpu() - K.inverse()).abs().max() < 1e-6


@pytest.mark.parametrize("math", [hip.MATH_F32, hip.MATH_BF16X3], ids=["f32", "bf16x3"])
def test_splitk_fixup_sees_fresh_partials(hiplib, math):
    """The split-K fix-up reads the other slices' partial sums across XCD-private L2s.  Re-launch ONE split-K conv with a
    new input every time (many tiles, so the slices of a tile land on different XCDs): a stale partial from the previous
    launch would give the previous answer.  Also checks the arrival counters are zero again after every launch."""
    from dd3d_amd.engine import ConvOp, PlanBase, pack_filter
    B, H, W, Cin, Cout = 1, 24, 80, 256, 256
    g = torch.Generator().manual_seed(11)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9)**0.5
    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    wp, meta = pack_filter(w, plan.device)
    xin, yout = plan.buf("x", B, H, W, Cin), plan.buf("y", B, H, W, Cout)
    ones, zeros = torch.ones(Cout, device=plan.device), torch.zeros(Cout, device=plan.device)
    seg = {"in": xin.view(), "out": yout.view(), "w": wp, "scale": ones, "bias": zeros}
    for tile, sk in ((hip.TILE_128x64, 4), (hip.TILE_64x64, 8) if math == hip.MATH_F32 else (hip.TILE_256x128, 3)):
        op = ConvOp(plan, meta, 1, 1, [seg], False, tile=tile, splitk=sk, name="stress", math=math)
        wd = w.to(plan.device)
        for it in range(12):
            x = torch.randn(B, Cin, H, W, generator=g)
            xin.t.copy_(x.permute(0, 2, 3, 1))
            op(plan.lib, hip.current_stream())
            torch.cuda.synchronize()
            ref = F.conv2d(x, w, None, padding=1)
            got = yout.nchw().cpu()
            assert float((got - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), (tile, sk, it)
            assert int(op.counters.abs().sum()) == 0


def test_split_bf16_is_f32_accurate(hiplib):
    """Error of both arithmetic modes against a float64 convolution of the same f32 data: the split-operand (3 x bf16, 6
    products) path must sit at f32 rounding level like the f32-MFMA path -- not at bf16 level (which would be ~1e-2)."""
    from dd3d_amd.engine import ConvOp, PlanBase, pack_filter
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 1, 24, 40, 256, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 48.0
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    wp, meta = pack_filter(w, plan.device)
    xin, yout = plan.buf("x", B, H, W, Cin), plan.buf("y", B, H, W, Cout)
    xin.t.copy_(x.permute(0, 2, 3, 1))
    ones, zeros = torch.ones(Cout, device=plan.device), torch.zeros(Cout, device=plan.device)
    seg = {"in": xin.view(), "out": yout.view(), "w": wp, "scale": ones, "bias": zeros}
    err = {}
    for math in (hip.MATH_F32, hip.MATH_BF16X3):
        op = ConvOp(plan, meta, 1, 1, [seg], False, name="acc", math=math)
        op(plan.lib, hip.current_stream())
        torch.cuda.synchronize()
        err[math] = float((yout.nchw().cpu().double() - ref).abs().max() / ref.abs().max())
    print("max |err| / max |ref|: f32 MFMA %.2e, split bf16 %.2e" % (err[hip.MATH_F32], err[hip.MATH_BF16X3]))
    assert err[hip.MATH_F32] < 2e-6 and err[hip.MATH_BF16X3] < 2e-6
    assert err[hip.MATH_BF16X3] < 3 * err[hip.MATH_F32] + 2e-7


@pytest.mark.parametrize("case", [
    ("base7x7_c3", 2, 43, 75, 3, 16, 7, 1, 3), ("level0_c16", 1, 40, 72, 16, 16, 3, 1, 1), ("level1_c16_s2", 1, 41, 73, 16, 32, 3, 2, 1),
    ("v99_stem1_c3_s2", 1, 40, 72, 3, 64, 3, 2, 1), ("base7x7_big", 1, 96, 200, 3, 16, 7, 1, 3)
], ids=lambda c: c[0])
def test_smallc_patch_conv_matches_torch(hiplib, case):
    """The stem kernel (input patch in LDS, split-bf16 MFMA 16x16x32, no im2col loop) vs F.conv2d, odd sizes included."""
    from dd3d_amd.engine import PlanBase, SmallcConvOp
    name, B, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k)**0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    Ho, Wo = ref.shape[-2:]
    plan = PlanBase("cuda")
    plan.math = hip.MATH_BF16X3  # these tests drive the f32-input kernels (f32 MFMA / on-the-fly bf16x3 split); the split-plane path: test_conv_planes_gpu.py
    cin_p = 4 if Cin <= 4 else 16
    assert plan.lib.dd3d_conv2d_smallc_supported(cin_p, k, k, stride, pad, Cout)
    xin = plan.buf("x", B, H, W, cin_p)
    xin.t[..., :Cin] = x.permute(0, 2, 3, 1).to(plan.device)
    yout = plan.buf("y", B, Ho, Wo, Cout + 4)
    yout.t.fill_(-777.0)
    op = SmallcConvOp(plan, w, cin_p, stride, pad, xin.view(), yout.view(0, Cout), scale.to(plan.device), bias.to(plan.device), True, name)
    op(plan.lib, hip.current_stream())
    torch.cuda.synchronize()
    got = yout.t[..., :Cout].permute(0, 3, 1, 2).cpu()
    err = float((got - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (name, err)
    assert torch.all(yout.t[..., Cout:] == -777.0)
