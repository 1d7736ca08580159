"""Lowering of the backbones and the FPN onto launch plans: DLA (dla.py:170-355), VoVNet-V2 (vovnet.py:218-238,357-367), detectron2's FPN
+ LastLevelP6P7 / LastLevelP6 [ext] -- a mixin of ForwardPlan (dd3d_amd.engine.forward)."""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm

from dd3d_amd.engine.ops import CallOp, ConvOp, FusedStemOp
from dd3d_amd.engine.packing import dense_filter, pad32, scatter_in_channels


class BackboneLowering:
    """Methods that append the backbone / FPN launches of a model to a PlanBase (self)."""
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
            stem = FusedStemOp(self, self.model, [dla.base_layer, dla.level0[0], dla.level1[0]], y.view(), name="stem")
            stem.begin_forward = bool(getattr(self, "stem_begins_forward", False))
            self.ops.append(stem)
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

