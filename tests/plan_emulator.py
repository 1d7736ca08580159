"""CPU emulator of a (dry-run) launch plan -- TEST INFRASTRUCTURE.  Executes the ops of dd3d_amd.engine.ForwardPlan with plain torch
on the plan's own (CPU) buffers, reading what each op computes from its `desc`: packed filters, folded norm / Scale / Offset vectors,
channel-slice views, concat-by-placement, residual sources.  What it checks is the HOST side of the engine (the part that changes with
the configuration); the kernels' arithmetic is checked on the GPU.  Stops at the first op it has no description for (select/decode)."""
import torch
import torch.nn.functional as F


def unpack_filter(wp, meta):
    """Inverse of dd3d_amd.engine.pack_filter: Wp[Npad][Kpad], k = (c/CC)*(T*CC) + tap*CC + c%CC  ->  OIHW [N][Cin][KH][KW]."""
    N, Cin, KH, KW = meta["N"], meta["Cin"], meta["KH"], meta["KW"]
    CC, T = min(Cin, 32), KH * KW
    w = wp[:N, :T * Cin].reshape(N, Cin // CC, T, CC)  # [n][chunk][tap][c % CC]
    return w.permute(0, 1, 3, 2).reshape(N, Cin, KH, KW)


def _store(view, y_nchw, n):
    view.buf.t[..., view.c0:view.c0 + n] = y_nchw.permute(0, 2, 3, 1)


def _conv(d):
    meta = d["meta"]
    for s in d["segs"]:
        x = s["in"].nchw()
        if d["in_relu"]:
            x = F.relu(x)
        n = int(s.get("n_limit") or meta["N"])
        w = unpack_filter(s["w"], dict(meta, N=n))
        acc = F.conv2d(x, w, None, stride=d["stride"], padding=d["pad"])
        y = acc * s["scale"][:n].view(1, -1, 1, 1) + s["bias"][:n].view(1, -1, 1, 1)
        if s.get("res") is not None:
            r = s["res"].buf.t[..., s["res"].c0:s["res"].c0 + n].permute(0, 3, 1, 2)
            if s.get("res_up"):  # the FPN top-down sum fused into the lateral convolution (dd3d_conv_seg.res_mode 3)
                r = F.interpolate(r, scale_factor=2, mode="nearest")
            y = y + r
        lo = torch.full((n, ), float("-inf")) if s.get("lo") is None else s["lo"][:n].clone()
        if d["relu"]:
            lo = lo.clamp(min=0.0)
        _store(s["out"], torch.maximum(y, lo.view(1, -1, 1, 1)), n)


def _smallc(d):
    w = d["weight"].detach().float()
    x = d["vin"].nchw()[:, :w.shape[1]]
    n = w.shape[0]
    y = F.conv2d(x, w, None, stride=d["stride"], padding=d["pad"]) * d["scale"][:n].view(1, -1, 1, 1) + d["bias"][:n].view(1, -1, 1, 1)
    _store(d["vout"], F.relu(y) if d["relu"] else y, n)


def _fused_stem(plan, d):
    """csrc/stem_fused.hip: the normalised canvas (the preprocess op's buffer) through base_layer -> level0 -> level1, each + folded
    norm + ReLU."""
    x = plan.bufs["img4"].nchw(0, 3)
    for cv in d["convs"]:
        w = cv["weight"].detach().float()
        n = w.shape[0]
        x = F.relu(F.conv2d(x, w, None, stride=cv["stride"], padding=cv["pad"]) * cv["scale"][:n].view(1, -1, 1, 1) + cv["bias"][:n].view(1, -1, 1, 1))
    _store(d["vout"], x, x.shape[1])


def _preprocess(plan, d):
    img = d["img"].t
    img.zero_()
    mean, std = torch.tensor(d["mean"]), torch.tensor(d["std"])
    for b in range(plan.B):
        h, w = plan.in_sizes[b].tolist()
        img[b, :h, :w, :3] = (plan.in_u8[b, :, :h, :w].float().permute(1, 2, 0) - mean) / std
    plan.inv_K.copy_(torch.linalg.inv(plan.in_K.view(-1, 3, 3)).reshape(-1, 9))


class Forms:
    """Which storage (f32 NHWC / split planes) of which channels of every buffer has been written so far.  A dry-run buffer keeps ONE
    logical tensor for the emulation; on the device the two storages are separate memories, and a convolution reading the planes of a
    buffer only its f32 side was written to (or the other way round) would read stale zeros.  Every op's reads are checked here."""
    def __init__(self):
        self.done = {}

    def wrote(self, view, form, n=None):
        has = view.buf.np if form == "planes" else view.buf.has_f32
        assert has, f"{view.buf.name}: op writes the {form} storage, which this buffer does not have"
        self.done.setdefault((id(view.buf), form), set()).update(range(view.c0, view.c0 + (view.C if n is None else n)))

    def reads(self, view, form, what, n=None):
        got = self.done.get((id(view.buf), form), set())
        need = set(range(view.c0, view.c0 + (view.C if n is None else n)))
        # channels nobody ever writes (32-channel padding of a concat slice) are zeros in both storages from allocation
        other = self.done.get((id(view.buf), "planes" if form == "f32" else "f32"), set())
        missing = (need - got) & other
        assert not missing, f"{what}: reads the {form} storage of {view.buf.name} channels {min(missing)}..{max(missing)}, which only hold the other form"


def _track(forms, name, d):
    k = d["kind"]
    if k == "conv":
        for sg, (wf, wp), rform in zip(d["segs"], d["out_forms"], d["res_forms"]):
            forms.reads(sg["in"], d["in_form"], name)
            if sg.get("res") is not None:
                assert rform in ("f32", "planes", "planes_up") and (rform == "planes_up") == bool(sg.get("res_up")), (name, rform)
                forms.reads(sg["res"], "f32" if rform == "f32" else "planes", name + " (residual)", d["meta"]["N"])
            n = int(sg.get("n_limit") or d["meta"]["N"])
            if wf:
                forms.wrote(sg["out"], "f32", n)
            if wp:
                forms.wrote(sg["out"], "planes", (n + 31) // 32 * 32)
    elif k == "smallc_conv":
        forms.reads(d["vin"], "f32", name)
        forms.wrote(d["vout"], "f32", d["weight"].shape[0])
    elif k == "preprocess":
        forms.wrote(d["img"].view(), "f32")
    elif k == "fused_stem":
        if d["vout"].buf.has_f32:
            forms.wrote(d["vout"], "f32")
        if d["planes"]:
            forms.wrote(d["vout"], "planes")
    elif k == "split_planes":
        forms.reads(d["src"], "f32", name)
        forms.wrote(d["dst"], "planes")
    elif k in ("maxpool2x2", "maxpool3x3s2_ceil"):
        forms.reads(d["vin"], d.get("in_form", "f32"), name)
        if d["vout"].buf.has_f32 and d.get("in_form", "f32") == "f32":
            forms.wrote(d["vout"], "f32")
        if d.get("planes"):
            forms.wrote(d["vout"], "planes")
    elif k == "upsample2x_add":
        forms.reads(d["fine"], "f32", name)
        forms.reads(d["coarse"], "f32", name)
        forms.wrote(d["fine"], "f32")
        if d.get("planes"):
            forms.wrote(d["fine"], "planes")
    elif k == "ese":
        forms.reads(d["x"], "f32", name)
        if d["identity"] is not None:
            forms.reads(d["identity"], "f32", name)
        forms.wrote(d["out"], "f32")
        if d.get("planes"):
            forms.wrote(d["out"], "planes")


def emulate(plan, stop_before=("select_decode", )):
    """Run the plan's ops in order on its CPU buffers; returns the names of the ops executed."""
    done = []
    forms = Forms()
    for op in plan.ops:
        if op.name in stop_before:
            break
        d = getattr(op, "desc", None)
        if d is None:
            raise NotImplementedError(f"op {op.name!r} carries no description")
        k = d["kind"]
        _track(forms, op.name, d)
        if k == "conv":
            _conv(d)
        elif k == "smallc_conv":
            _smallc(d)
        elif k == "preprocess":
            _preprocess(plan, d)
        elif k == "fused_stem":
            _fused_stem(plan, d)
        elif k == "split_planes":
            if d["dst"] is not d["src"]:
                x = d["src"].nchw()
                _store(d["dst"], F.relu(x) if d["relu"] else x, d["src"].C)
            else:
                assert not d["relu"]
        elif k == "maxpool2x2":
            _store(d["vout"], F.max_pool2d(d["vin"].nchw(), 2), d["vin"].C)
        elif k == "maxpool3x3s2_ceil":
            _store(d["vout"], F.max_pool2d(d["vin"].nchw(), 3, 2, ceil_mode=True), d["vin"].C)
        elif k == "upsample2x_add":
            f, c = d["fine"], d["coarse"]
            _store(f, f.nchw() + F.interpolate(c.nchw(), scale_factor=2, mode="nearest"), f.C)
        elif k == "ese":
            x = d["x"].nchw()
            gate = F.relu6(F.linear(x.mean((2, 3)), d["weight"].view(x.shape[1], x.shape[1]), d["bias"]) + 3.0) / 6.0
            y = x * gate[:, :, None, None]
            if d["identity"] is not None:
                y = y + d["identity"].nchw()
            _store(d["out"], y, x.shape[1])
        elif k == "aligned_bilinear_scale":
            from oracle.dense_depth_oracle import aligned_bilinear
            y = aligned_bilinear(d["src"].nchw(0, 1), d["factor"], "half" if d["offset_half"] else "none")[:, 0]
            if d["focal_factor"] > 0:  # dense_depth.py:147-150
                inv_K = plan.inv_K.view(-1, 3, 3)
                pixel_size = torch.sqrt(inv_K[:, 0, 0]**2 + inv_K[:, 1, 1]**2)
                y = y / (pixel_size * d["focal_factor"]).view(-1, 1, 1)
            d["out"].copy_(y)
        else:
            raise NotImplementedError(k)
        done.append(op.name)
    return done
