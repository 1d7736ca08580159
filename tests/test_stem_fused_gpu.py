"""The one-launch DLA stem (csrc/stem_fused.hip, C ABI dd3d_stem_fused_f16x2) against a plain PyTorch fp32 statement of the same four
steps -- (x - mean) / std with ImageList's zero padding, base_layer 7x7, level0 3x3, level1 3x3 stride 2, each + folded norm + ReLU
(dla.py:271-280,327-344) -- and against the launch-by-launch lowering it replaces.  Tolerance: the f32 kernels' (2e-5 of the largest
entry); the plane output must be the two-half-term split of the f32 output."""
import pytest
import torch
import torch.nn.functional as F

from dd3d_amd import hip

pytestmark = pytest.mark.gpu


def _stem_case(B, Hp, Wp, sizes, seed=0):
    from dd3d_amd.engine import FusedStemOp, PlanBase
    from dd3d_amd.layers import fold_norm
    from tests.util import bundle, gpu_model
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti")
    model = gpu_model(cfg, sd, use_graph=False)
    dla = model.backbone.bottom_up
    plan = PlanBase("cuda")
    plan.math = hip.MATH_F16X2
    plan.B, plan.Hp, plan.Wp = B, Hp, Wp
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 3, Hp, Wp), dtype=torch.uint8, generator=g)
    plan.in_u8 = u8.cuda()
    plan.in_sizes = torch.tensor(sizes, dtype=torch.int32).cuda()
    out = plan.buf("level1", B, Hp // 2, Wp // 2, 32 + 32, kind="both")  # a channel slice of a wider buffer
    out.t.fill_(-777.0)
    out.p.fill_(0x1234)
    convs = [dla.base_layer, dla.level0[0], dla.level1[0]]
    op = FusedStemOp(plan, model, convs, out.view(32, 32))
    # reference, float64 accumulation where torch allows (CPU double conv)
    mean, std = model.pixel_mean.cpu().view(1, 3, 1, 1), model.pixel_std.cpu().view(1, 3, 1, 1)
    x = torch.zeros(B, 3, Hp, Wp)
    for b, (h, w) in enumerate(sizes):
        x[b, :, :h, :w] = ((u8[b, :, :h, :w].float() - mean[0]) / std[0])
    ref = x.double()
    for cv in convs:
        sc, sh = fold_norm(cv, None)
        ref = F.relu(F.conv2d(ref, cv.weight.detach().cpu().double(), None, stride=cv.stride, padding=cv.padding) * sc.cpu().double().view(1, -1, 1, 1)
                     + sh.cpu().double().view(1, -1, 1, 1))
    return plan, op, out, ref.float(), model, (u8, sizes)


@pytest.mark.parametrize("B,Hp,Wp,sizes", [
    (1, 128, 256, [(128, 256)]),
    (2, 128, 256, [(128, 256), (115, 234)]),       # ragged batch: the second image is zero-padded after normalisation
    (1, 52, 76, [(50, 71)]),                       # level1 map 26 x 38: partial tiles in both directions, every tile touches a border
    (3, 16, 64, [(16, 64), (9, 33), (16, 1)]),     # one row of tiles; a one-pixel-wide image
], ids=["128x256", "ragged_b2", "partial_tiles", "tiny_b3"])
def test_fused_stem_matches_torch(hiplib, B, Hp, Wp, sizes):
    plan, op, out, ref, _, _ = _stem_case(B, Hp, Wp, sizes)
    plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    got = out.t[..., 32:64].permute(0, 3, 1, 2).cpu()
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), tol)
    assert torch.all(out.t[..., :32] == -777.0)  # the other channels of the wider buffer are untouched
    assert torch.all(out.p[0] == 0x1234)
    terms = out.p[1].view(torch.float16).float() / out.plane_scale  # [B*Ho*Wo][2][32]
    dec = terms.sum(1).view(B, Hp // 2, Wp // 2, 32).permute(0, 3, 1, 2).cpu()
    assert float((dec - got).abs().max()) <= 2.0**-21 * float(got.abs().max()) + 2.0**-24 / out.plane_scale
    assert int(plan.status.cpu()) == 0
    plan.launch()  # idempotent
    torch.cuda.synchronize()
    assert torch.equal(out.t[..., 32:64].permute(0, 3, 1, 2).cpu(), got)


def test_fused_stem_equals_the_launch_by_launch_stem_in_the_forward(hiplib, monkeypatch):
    """Whole forward with and without the fused stem: same detections; level1 maps within the conv tolerance."""
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle, gpu_model, max_abs
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti")
    inputs = make_inputs(2, 128, 256)
    inputs[1]["image"] = inputs[1]["image"][:, :117, :231].contiguous()
    fused = gpu_model(cfg, sd, use_graph=False)
    pf, sizes = fused.stage_inputs(inputs)
    pf.run()
    of = fused.collect(pf, inputs, sizes)
    assert pf.fused_stem and pf.stem_begins_forward and [op.name for op in pf.ops[:2]] == ["stem", "level2.pool"]  # no separate start-of-forward launches
    monkeypatch.setenv("DD3D_FUSED_STEM", "0")
    plain = gpu_model(cfg, sd, use_graph=False)
    pp, _ = plain.stage_inputs(inputs)
    pp.run()
    op_ = plain.collect(pp, inputs, sizes)
    assert not pp.fused_stem and "base_layer" in [op.name for op in pp.ops]
    a, b = pf.bufs["level1.0"].nchw(), pp.bufs["level1.0"].nchw()
    assert max_abs(a, b) <= 2e-5 * float(b.abs().max())
    assert torch.equal(pf.normalized_image(), pp.normalized_image())
    for x, y in zip(of, op_):
        x, y = x["instances"], y["instances"]
        assert len(x) == len(y) > 0 and torch.equal(x.pred_classes, y.pred_classes) and torch.equal(x.locations, y.locations)
        assert torch.allclose(x.pred_boxes.tensor, y.pred_boxes.tensor, rtol=1e-4, atol=1e-3) and torch.allclose(x.scores_3d, y.scores_3d, rtol=1e-4)


def test_fused_stem_flags_values_outside_the_half_range(hiplib):
    """A plane scale that pushes the stem's activations past 65504 trips the status word (the forward then raises / falls back)."""
    plan, op, out, ref, _, _ = _stem_case(1, 32, 64, [(32, 64)])
    op.a.plane_scale = 65504.0 * 4 / max(float(ref.abs().max()), 1e-3)  # (the epilogue scales were folded for the old scale: values only need to overflow)
    plan.ops.append(op)
    plan.launch()
    torch.cuda.synchronize()
    with pytest.raises(FloatingPointError, match="half range"):
        plan.check_status()
