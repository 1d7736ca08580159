"""Launch-plan operations: one object per libdd3d_hip launch (dd3d_amd.engine)."""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm

from dd3d_amd.engine.packing import pack_smallc_bf16x3, pack_smallc_f16x2
from dd3d_amd.engine.tiling import MATH_TILES, PLANE_TILES, ROW_ONLY_TILES, choose_tiling, preferred_tile


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
        # start-of-forward chores the launch takes along (no separate launches): K^-1 and the per-forward range-guard maxima.  Their
        # addresses / extents are only final once the whole plan is built (the record takes over inv_K, convolutions register their
        # maxima): `begin_forward` = True makes __call__ read them off the plan at launch (= capture) time.
        self.plan, self.begin_forward = plan, False
        self.a = a
        self.desc = dict(kind="fused_stem", convs=descs, vout=vout, mean=[float(v) for v in model.pixel_mean.flatten()],
                         std=[float(v) for v in model.pixel_std.flatten()], planes=bool(vout.np))
        self.info = dict(name=name, M=B * (Hp // 2) * (Wp // 2), N=32, K=0, tile="fused", splitk=1, math=hip.MATH_F16X2, blocks=0, nsegs=1)

    def __call__(self, lib, stream):
        if self.begin_forward:
            p, a = self.plan, self.a
            a.K, a.inv_K = p.in_K.data_ptr(), p.inv_K.data_ptr()
            n = max(1, len(p.amax_names))
            a.zero_f32, a.zero_count = p.amax.data_ptr(), n * p.amax.shape[1] * p.amax.shape[2]
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


class ConvOp:
    """One dd3d_conv2d_igemm_f32 launch (possibly many segments).  Input form: the split planes of the input buffers when they have
    them (plan.use_planes), else f32 NHWC.  Output form per segment: every storage its output buffer has (f32 NHWC and / or split
    planes), unless the segment says `write_f32=False` / `write_planes=False`."""
    def __init__(self, plan, meta, stride, pad, segs, relu, tile=None, splitk=None, name="", math=None, in_relu=False, chain=False):
        """`chain`: the segments DEPEND on each other (include/dd3d_hip.h, dd3d_conv_launch.chain): a segment whose dict carries
        "dep" = index of an earlier segment reads that segment's output as its input; residuals may come from any earlier segment."""
        dev = plan.device
        self.name = name
        self.chain = bool(chain)
        m_list = [s["out"].B * s["out"].H * s["out"].W for s in segs]
        if math is None:
            math = plan.math
        in_planes = math != hip.MATH_F32 and meta["Cin"] % 32 == 0 and all(s["in"].np == hip.MATH_PLANES[math] for s in segs) and not in_relu
        if meta["Cin"] % 32 or (meta["N"] <= 32 and not in_planes):  # stem layers (Cin 4 / 16) and narrow convs on f32 input: the f32 kernel
            math = hip.MATH_F32
        if math in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2) and not in_planes:
            raise ValueError(f"conv {name}: math mode {math} reads split-plane input only; its input buffer has none")
        self.math, self.in_planes = math, in_planes
        cfg, sk = choose_tiling(m_list, meta["N"], meta["Kpad"], stride, math, planes=in_planes, policy=getattr(plan, "tile_policy", "latency"))
        if tile is not None:
            if tile not in (PLANE_TILES if in_planes else MATH_TILES[math]):
                raise ValueError(f"conv {name}: tile {hip.TILE_NAMES[tile]} is not instantiated for math mode {math}")
            cfg = tile
            pref = preferred_tile(m_list, meta["N"], meta["Kpad"], stride, math, planes=in_planes, policy=getattr(plan, "tile_policy", "latency"))
            if pref is not None and splitk is None:  # a sweep's override / the throughput table names this exact shape
                cfg, sk = pref
        if splitk is not None:
            sk = splitk
        if cfg in ROW_ONLY_TILES and not ((meta["KH"], meta["KW"], stride, pad) == (3, 3, 1, 1) and in_planes and sk == 1):
            cfg = hip.TILE_256x256_W8  # (instantiated for the row-shared 3 x 3 kernel only)
        if cfg in (hip.TILE_256x256_W8, hip.TILE_192x256_W8) and any(sg.get("res") is not None for sg in segs):
            cfg = hip.TILE_256x128  # (the 8-wave 256-column tiles have no registers left for a residual in flight)
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
            if chain:
                dep = s.get("dep")
                assert dep is None or (0 <= dep < i and segs[dep]["out"].c0 == vin.c0 and segs[dep]["out"].C == vin.C and segs[dep]["out"].buf is vin.buf), (name, i, dep)
                assert in_planes and wp and not wf and res_form in (None, "planes") and (stride, pad, meta["KH"], meta["KW"]) == (1, 1, 3, 3), (name, i)
                a["reserved"] = 0 if dep is None else dep + 1
                self.seg_tile0 = getattr(self, "seg_tile0", []) + [len(tiles)]
            tiles += [(i, m0) for m0 in range(0, m_list[i], bm)]
            self.keep += [w, scale_vec, s["bias"], s.get("lo")]  # (what the launch reads; kept alive here)
        self.desc = dict(kind="conv", segs=segs, meta=meta, stride=stride, pad=pad, relu=bool(relu), in_relu=bool(in_relu),
                         in_form="planes" if in_planes else "f32", out_forms=self.out_forms, res_forms=self.res_forms, chain=self.chain)
        self.ctor = dict(meta=meta, stride=stride, pad=pad, relu=bool(relu), tile=cfg, splitk=sk, math=math, in_relu=bool(in_relu))  # (merge_chains)
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
        self.chain_sync = self.chain_tile0 = None
        if chain:
            assert len(segs) >= 2 and not (cfg in (hip.TILE_256x256_W8, hip.TILE_192x256_W8) and sk > 1)
            self.chain_sync = torch.zeros(1 + len(tiles), dtype=torch.int32, device=dev)  # zero between launches (the last tile clears it)
            self.chain_tile0 = torch.tensor(self.seg_tile0, dtype=torch.int32).to(dev)
            L.chain, L.chain_sync, L.chain_tile0 = 1, self.chain_sync.data_ptr(), self.chain_tile0.data_ptr()
        L.amax = plan.amax_slot(name) if (math == hip.MATH_F16X2 and any(wp for _, wp in self.out_forms) and not plan.dry_run
                                          and os.environ.get("DD3D_AMAX", "1") != "0") else None  # DD3D_AMAX=0: A/B measurements only
        self.L = L
        # algorithmic MACs: every segment counts the channels it stores
        # (a segment that repeats another's products -- relu(p6) beside p6 -- is marked `algorithmic=False` and not counted)
        self.macs = sum(m * (sg.get("n_limit") or meta["N"]) for m, sg in zip(m_list, segs) if sg.get("algorithmic", True)) * meta["KH"] * meta["KW"] * meta["Cin"]
        self.info = dict(name=name, M=sum(m_list), N=meta["N"], K=meta["Kpad"], tile=(bm, bn), tile_name=hip.TILE_NAMES[cfg], splitk=sk, math=math,
                         blocks=len(tiles) * -(-meta["N"] // bn) * sk, nsegs=len(segs), in_form="planes" if in_planes else "f32", chain=self.chain)

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

