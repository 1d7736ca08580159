"""The two-launch eSE module (dd3d_ese_fused: pool + per-image mean, then gate + scale (+ identity) -> f32 and split planes) against a
plain PyTorch fp32 statement of vovnet.py:180-185,248-249 and against the three-launch dd3d_ese_nhwc."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# B, H, W, buffer channels, real channels, identity, channel offset of the output view inside a wider buffer
CASES = [(1, 12, 40, 1024, 1024, True, 0), (2, 24, 80, 768, 768, False, 0), (1, 96, 320, 256, 256, False, 0), (3, 7, 9, 96, 80, True, 32),
         (2, 48, 160, 512, 512, True, 64), (1, 1, 1, 32, 32, False, 0), (16, 12, 40, 1024, 1024, True, 0)]


def _reference(x, identity, fc):
    y = x.mean((2, 3), keepdim=True)
    y = F.relu6(F.conv2d(y, fc.weight, fc.bias) + 3.0) / 6.0
    out = x * y
    return out + identity if identity is not None else out


@pytest.mark.parametrize("math", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("B,H,W,Cb,Cr,ident,c0", CASES)
def test_fused_ese_matches_reference_and_the_three_launch_entry(hiplib, math, B, H, W, Cb, Cr, ident, c0):
    from dd3d_amd import hip
    from dd3d_amd.engine import MATH_NAMES, PlanBase
    gen = torch.Generator().manual_seed(B * 1000 + H)
    plan = PlanBase("cuda")
    plan.math = MATH_NAMES[math]
    x = torch.randn(B, Cr, H, W, generator=gen) * 1.5
    idn = torch.randn(B, Cr, H, W, generator=gen) if ident else None
    fc = torch.nn.Conv2d(Cr, Cr, 1)
    with torch.no_grad():
        fc.weight.copy_(torch.randn(Cr, Cr, 1, 1, generator=gen) / Cr**0.5)
        fc.bias.copy_(torch.randn(Cr, generator=gen))
    kind = "both" if plan.use_planes else "f32"
    xb, ib, ob = plan.buf("x", B, H, W, Cb), plan.buf("id", B, H, W, Cb), plan.buf("out", B, H, W, c0 + Cb, kind=kind)
    xb.t[..., :Cr].copy_(x.permute(0, 2, 3, 1))
    if ident:
        ib.t[..., :Cr].copy_(idn.permute(0, 2, 3, 1))
    out = ob.view(c0, Cb)
    plan.ese(xb.view(), ib.view() if ident else None, out, fc, name="ese")
    assert [op.name for op in plan.ops] == ["ese"]  # no separate split of the result
    with torch.no_grad():
        want = _reference(x, idn, fc)
    for rep in range(2):  # the arrival counters are left zero: a second launch behaves like the first
        ob.t.zero_()
        plan.launch()
        torch.cuda.synchronize()
        plan.check_status()
        got = out.nchw()[:, :Cr].cpu()
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), (rep, float((got - want).abs().max()))
        assert plan.ops[0].keep[4].cpu().tolist() == [0] * B
    assert float(out.nchw()[:, Cr:].abs().max()) == 0.0 if Cb > Cr else True  # padded channels stay zero
    if plan.use_planes:  # the planes written by the same launch encode the f32 result
        f32 = out.nchw().clone()
        keep_t, ob.t = ob.t, None
        try:
            dec = ob.nchw(c0, Cb)
        finally:
            ob.t = keep_t
        tol = 0.0 if math == "bf16x3" else 2.0**-20
        assert float((dec - f32).abs().max()) <= tol * max(1.0, float(f32.abs().max())), float((dec - f32).abs().max())
    # the three-launch entry point sums in the same order
    lib = hip.lib()
    op = plan.ops[0]
    w, b, partial, gate = op.keep[:4]
    ref_out = torch.zeros_like(ob.t)
    rs = partial.shape[1]
    scratch = [torch.zeros_like(partial), torch.zeros_like(gate)]  # kept alive across the launch
    hip.check(lib.dd3d_ese_nhwc(xb.view().ptr, ib.view().ptr if ident else None, ref_out.data_ptr() + 4 * c0, w.data_ptr(), b.data_ptr(), scratch[0].data_ptr(),
                                scratch[1].data_ptr(), B, H * W, Cb, xb.pitch, ib.pitch if ident else 0, ob.pitch, rs, hip.current_stream()), "ese_nhwc")
    torch.cuda.synchronize()
    assert torch.allclose(ref_out[..., c0:c0 + Cb], ob.t[..., c0:c0 + Cb], rtol=1e-6, atol=1e-7)


def test_fused_ese_rejects_bad_shapes(hiplib):
    from dd3d_amd import hip
    lib = hip.lib()
    t = torch.zeros(64, device="cuda")
    p = t.data_ptr()
    assert lib.dd3d_ese_fused(p, None, p, None, p, p, p, p, p, 1, 4, 4100, 4100, 0, 4100, 1, 0, 1.0, None, None) != 0
    assert "multiple of 8 up to 4096" in lib.dd3d_last_error().decode()
    assert lib.dd3d_ese_fused(p, None, None, p, p, p, p, p, p, 1, 4, 24, 24, 0, 24, 1, hip.MATH_F16X2, 1.0, None, None) != 0
    assert "C % 32" in lib.dd3d_last_error().decode()
