"""PlanBase: buffer / weight-store bookkeeping, op helpers, and the runner of a launch sequence -- launch by launch or as one captured
hipGraph (dd3d_amd.engine)."""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm

from dd3d_amd.engine.ops import CallOp, ConvOp, OpList, SmallcConvOp
from dd3d_amd.engine.packing import Buf, dense_filter, pack_filter, split_f16x2_host, split_planes_host
from dd3d_amd.engine.tiling import default_math


# --------------------------------------------------------------------------------------------- the plan
class HalfRangeOverflow(FloatingPointError):
    """DD3D_MATH_F16X2: an activation left the half format's range while being split (|x * plane scale| > 65504, or a NaN / inf)."""


class HalfRangeUnderflow(FloatingPointError):
    """DD3D_MATH_F16X2: a convolution's outputs sit below the useful part of the half pair's range (largest |x * plane scale| < 2^-5)."""


def relax_arithmetic(model, err):
    """What a model on the DEFAULT arithmetic (model.math is None) does after the f16x2 range guard fired -- shared by DD3D.forward,
    DD3DDenseDepth and the runners of dd3d_amd.parallel.  An OVERFLOW first widens the half range: a plane scale of 16 / 4 / 1 holds
    activations up to 4094 / 16376 / 65504 at the f16x2 speed (the pair's absolute floor rises with it; the underflow side of the guard
    keeps watching); only beyond that -- or on an underflow, which no smaller scale can cure -- the model moves to the three-term bf16
    split (full f32 exponent range, twice the matrix work).  The target scale is chosen in ONE step from the largest activation the guard
    sampled (`err.sampled_max_abs`, a lower bound of the true maximum: the scale must leave it a factor of two; round-5 advisor: stepping
    16 -> 4 -> 1 -> bf16x3 blindly cost a corrupted frame three plan rebuilds per pipeline slot), by factors of four down to 1; a
    NON-FINITE maximum (a NaN / inf input frame trips the overflow bit too) goes straight to bf16x3, as does a maximum no half can hold.
    Without a sample (the verdict came from another rank's record) the scale goes down one factor of four per call.  Returns False when
    nothing is left to relax (the caller re-raises): an arithmetic that was asked for explicitly is never changed.  With several ranks
    every rank reads the same verdict out of the exchanged records (ForwardPlan.check_status), so all of them rebuild on the same step."""
    import warnings
    if model.math is not None or default_math() != hip.MATH_F16X2:
        return False
    cur = float(model.act_scale) if getattr(model, "act_scale", None) else float(os.environ.get("DD3D_F16_ACT_SCALE", "16"))
    amax = getattr(err, "sampled_max_abs", None)
    finite = amax is None or math.isfinite(amax)
    new = None
    if isinstance(err, HalfRangeOverflow) and cur > 1.0 and finite:
        new = max(1.0, cur / 4.0)
        if amax:
            while new > 1.0 and amax * new >= PlanBase.HALF_MAX / 2.0:
                new = max(1.0, new / 4.0)
            if amax * new >= PlanBase.HALF_MAX:
                new = None  # no half holds it at any scale
    if new is not None:
        model.act_scale = new
        model._range_relaxations = int(getattr(model, "_range_relaxations", 0)) + 1
        warnings.warn(f"dd3d_amd: {err}; keeping f16x2 with the plane scale lowered {cur:g} -> {model.act_scale:g} "
                      f"(activations up to {65504.0 / model.act_scale:g}" + (f"; largest sampled |x| = {amax:.4g})" if amax else ")"))
    else:
        model.math = "bf16x3"
        warnings.warn(f"dd3d_amd: {err}; switching this model to math='bf16x3'" + ("" if finite else " (non-finite activations)"))
    model._plans.clear()
    return True


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
        # "latency" (one plan at a time: the measured per-launch table) or "throughput" (the plan shares the chip with other plans:
        # tiling.THROUGHPUT_TILE_TABLE first); ForwardPlan takes the runner's / the model's choice, DD3D_TILE_POLICY overrides
        self.tile_policy = "latency"
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

    # ------------------------------------------------------------------ host <-> device hand-over
    def host_buf(self, shape, dtype):
        """Host staging memory: pinned where there is a device (asynchronous copies), plain for dry-run plans."""
        pin = self.device.type == "cuda" and not self.dry_run
        return torch.zeros(shape, dtype=dtype, pin_memory=pin)

    def inputs_writable(self):
        """Call before rewriting the host mirrors of the per-image scalars: the previous forward's copy out of them must have run."""
        if getattr(self, "_inputs_event", None) is not None:
            self._inputs_event.synchronize()

    def flush_inputs(self):
        """Ship the per-image scalars `stage_inputs` wrote into the pinned host mirrors: ONE asynchronous copy for sizes + intrinsics, one
        for the resize targets (they live in the exchange record), two more for plans with BEV stages (poses, sample ids) -- per FORWARD,
        however many requests share the plan."""
        nb = not self.dry_run
        self.in_meta.copy_(self.host_meta, non_blocking=nb)
        self.in_outsize.copy_(self.host_outsize, non_blocking=nb)
        if getattr(self, "has_bev_inputs", False) and self.host_pose is not None:
            self.in_pose.copy_(self.host_pose, non_blocking=nb)
            if not getattr(self, "camera_sharded", False):
                self.in_group.copy_(self.host_group, non_blocking=nb)
        if nb:
            if self._inputs_event is None:
                self._inputs_event = torch.cuda.Event()
            self._inputs_event.record()

    def _ensure_readback(self):
        """Buffers of dd3d_pack_readback (allocated at the first launch, i.e. outside any capture: every runner warms a plan up first)."""
        if getattr(self, "_rb_dev", None) is not None:
            return
        det_count = getattr(self, "det_count", None)
        self._rb_G = int(det_count.numel()) if det_count is not None else 0
        self._rb_n = len(self.amax_names) if self.math == hip.MATH_F16X2 else 0
        gathered = getattr(self, "gathered", None)
        self._rb_nrec = int(self.world_size) if (gathered is not None and getattr(self, "exchange", False) and self.math == hip.MATH_F16X2) else 0
        words = 4 + self._rb_G + self._rb_n + 2 * self._rb_nrec
        self._rb_dev = torch.zeros(words, dtype=torch.int32, device=self.device)
        self._rb_host = self.host_buf(words, torch.int32)
        self._rb_event = torch.cuda.Event()
        self._rb_pending, self._rb_cache = False, None

    def _launch_readback(self, st):
        """Last launch of a forward: everything the host reads afterwards, packed for one copy (include/dd3d_hip.h, dd3d_pack_readback)."""
        flags = None
        if self._rb_nrec:
            flags = self.gathered.data_ptr() + 4 * self.flags_off
        hip.check(self.lib.dd3d_pack_readback(self.det_count.data_ptr() if self._rb_G else None, self._rb_G, self.status.data_ptr(),
                                              self.amax.data_ptr() if self._rb_n else None, self._rb_n, flags, self._rb_nrec,
                                              int(getattr(self, "record_len", 0)), self._rb_dev.data_ptr(), st), "pack_readback")

    def fetch(self):
        """Enqueue, behind the forward just issued on the current stream, the ONE asynchronous device-to-host copy of its read-back words
        (detection counts, status, range-guard maxima, exchanged verdicts) into pinned memory.  `readback()` waits for it."""
        if self.dry_run:
            return
        self._ensure_readback()
        self._rb_host.copy_(self._rb_dev, non_blocking=True)
        self._rb_event.record()
        self._rb_pending, self._rb_cache = True, None

    def readback(self):
        """Host view of the last forward's read-back words: waits for the copy `fetch()` enqueued (or, when the forward was issued without
        one -- a test replaying launches by hand -- packs and copies now, on the current stream)."""
        from types import SimpleNamespace
        if self.dry_run:  # (host-order tests on dry-run plans: the same words straight off the plan's host tensors)
            det_count = getattr(self, "det_count", None)
            gathered = getattr(self, "gathered", None)
            flags = torch.zeros((0, 2), dtype=torch.int32)
            if gathered is not None and getattr(self, "exchange", False) and self.math == hip.MATH_F16X2:
                flags = gathered.view(self.world_size, self.record_len)[:, self.flags_off:self.flags_off + 2].view(torch.int32).clone()
            return SimpleNamespace(status=int(self.status), counts=det_count.clone() if det_count is not None else torch.zeros(0, dtype=torch.int32),
                                   amax=self.amax_values() if (self.amax_names and self.math == hip.MATH_F16X2) else torch.zeros(0), flags=flags)
        self._ensure_readback()
        if self._rb_cache is not None:
            return self._rb_cache
        if not self._rb_pending:
            self._launch_readback(hip.current_stream())
            self.fetch()
        self._rb_event.synchronize()
        self._rb_pending = False
        w = self._rb_host
        G, n, nrec = self._rb_G, self._rb_n, self._rb_nrec
        assert (int(w[1]), int(w[2]), int(w[3])) == (G, n, nrec), "read-back record does not match the plan"
        self._rb_cache = SimpleNamespace(status=int(w[0]), counts=w[4:4 + G].clone(), amax=w[4 + G:4 + G + n].view(torch.float32).clone(),
                                         flags=w[4 + G + n:4 + G + n + 2 * nrec].view(nrec, 2).clone())
        return self._rb_cache

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

    def check_status(self, rb=None):
        """Raise if a kernel flagged a numeric fault (reads one int32 and the per-launch maxima from the device; call after the forward
        has been waited for; `rb`: the forward's `readback()` -- nothing is read from the device then).  DD3D_MATH_F16X2 keeps activations as two IEEE halves of value * plane scale: exact to 2^-24 relative
        between 2^-1 and 65504, with an ABSOLUTE floor of 2^-25 below.  Overflow is flagged per element (status bit); underflow per
        tensor: a convolution whose LARGEST output, scaled, stayed below 2^-5 has lost more than four of its 24 bits."""
        st = int(self.status.cpu()) if rb is None else rb.status
        mx = None
        if self.amax_names and (rb is None or rb.amax.numel() == len(self.amax_names)):
            mx = self.amax_values() if rb is None else rb.amax
        if st & hip.STATUS_CHAIN_TIMEOUT:
            self.status.zero_()
            raise RuntimeError("a block of a chain launch (dd3d_conv_launch.chain) gave up waiting for its producer tiles: the results of this forward are "
                               "invalid.  Set DD3D_CHAIN=0 (one launch per convolution) and report this: the launch relies on workgroups being "
                               "dispatched in index order")
        if st & hip.STATUS_F16_OVERFLOW:
            self.status.zero_()
            e = HalfRangeOverflow(
                f"an activation left the half range while being split (|x| > {65504.0 / self.act_scale:g} at plane scale {self.act_scale:g}, or a "
                "NaN / inf): lower DD3D_F16_ACT_SCALE or run this model with math='bf16x3'")
            e.sampled_max_abs = self._sampled_max_abs(mx)  # (what relax_arithmetic sizes the next plane scale by)
            raise e
        if mx is not None:
            low = [(n, float(v)) for n, v in zip(self.amax_names, mx.tolist()) if 0.0 < v < self.AMAX_FLOOR]
            if low:
                n, v = min(low, key=lambda t: t[1])
                raise HalfRangeUnderflow(
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
                              f"overflow at {self.HALF_MAX / self.act_scale:g}); beyond it a model on the default arithmetic lowers its plane scale (16 -> 4 -> 1) and, "
                              "past 65504, falls back to bf16x3 at half the throughput")

    def _sampled_max_abs(self, mx):
        """Largest sampled |activation| (value units) behind an overflow verdict, from the per-launch maxima `mx` (launch order): the FIRST
        launch whose sampled maximum left the half range is the culprit -- everything downstream of it computed on infinities, so later
        maxima say nothing.  inf when the culprit's own maximum is not finite (non-finite INPUT); when no sample caught the overflowing
        element, the largest finite sample (a lower bound); None without samples."""
        if mx is None or not mx.numel():
            return None
        vals = [float(v) for v in mx.tolist()]
        for v in vals:
            if not (v < self.HALF_MAX):  # (>= the largest half, or NaN)
                return v / float(self.act_scale) if math.isfinite(v) else float("inf")
        top = max(vals)
        return top / float(self.act_scale) if top > 0.0 else None

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

    # ------------------------------------------------------------------ dependent convolutions in one launch
    def merge_chains(self, first=0):
        """Fuse runs of CONSECUTIVE convolution launches that feed each other into chain launches (include/dd3d_hip.h,
        dd3d_conv_launch.chain; csrc/conv_planes_row.hip): op[i + 1] reads op[i]'s output, same filter geometry / tile / split-K / flags,
        3 x 3 / stride 1 / pad 1 on split planes, planes-only outputs, residuals (if any) read from planes.  That is: conv2 of a DLA block
        with the following block's conv1 -> conv2 (dla.py:50-62, 233-247) -- the roots (1 x 1) and the strided first convolutions stay
        launches of their own -- and the four layers of the head towers (fcos2d.py:137-152, fcos3d.py:163-180), whose later layers then
        start on the CUs the previous layer's last round leaves idle.

        OFF by default (DD3D_CHAIN=1 turns it on, =backbone / =towers one of the two): bit-exact against the one-launch-per-convolution plan
        (tests/test_chain_gpu.py, also with five slots in flight), but MEASURED SLOWER on MI355X -- a dependent layer inside a launch costs
        5-9 us (write-through acknowledgement + arrival atomic + the consumer's poll + its first activation fetch past the L2) where a
        kernel boundary costs 1.5-2 us: one image 1.18 -> 1.27 ms, the driver command 1621 -> 1569 img/s, the four tower layers as one
        launch 1280 -> 1294 us per four images (the tail-round fill is eaten by the activations' three filter-row passes no longer hitting
        the L2): profiles/r06e_chain_prefix_*.txt, r06d_chain_ab.txt; DESIGN section 4, round 6."""
        mode = os.environ.get("DD3D_CHAIN", "0").strip().lower()
        if mode in ("0", "off") or not self.use_planes or os.environ.get("DD3D_CONV_ROW", "1") == "0":
            return
        ops = self.ops

        def chainable(op):
            if not isinstance(op, ConvOp) or op.chain or op.branch != 0 or op.joins:
                return False
            c = op.ctor
            m = c["meta"]
            if (m["KH"], m["KW"], c["stride"], c["pad"]) != (3, 3, 1, 1) or not op.in_planes or c["in_relu"] or c["math"] == hip.MATH_F32:
                return False
            if c["tile"] in (hip.TILE_256x256_W8, hip.TILE_192x256_W8) and c["splitk"] > 1:
                return False
            if c["splitk"] > 1 and -(-(m["Kpad"] // 32) // c["splitk"]) % 3:
                return False  # (K slices that do not start on a filter row run on the per-tap kernel, which has no chain form)
            tower = op.name.startswith("towers.")
            if (mode == "backbone" and tower) or (mode == "towers" and not tower):
                return False
            ok_out = all(wp and not wf for wf, wp in op.out_forms)
            ok_res = all(r in (None, "planes") for r in op.res_forms)
            return ok_out and ok_res and not any(sg.get("n_limit") for sg in op.desc["segs"])

        def feeds(prev, op):
            """segment k of `op` reads segment k of `prev` (same count), or a one-segment op reads the one-segment prev"""
            a, b = prev.desc["segs"], op.desc["segs"]
            if len(a) != len(b):
                return False
            return all(sb["in"].buf is sa["out"].buf and sb["in"].c0 == sa["out"].c0 and sb["in"].C == sa["out"].C for sa, sb in zip(a, b))

        def same_launch(x, y):
            cx, cy = x.ctor, y.ctor
            keys = ("N", "Cin", "KH", "KW", "Kpad", "Npad")
            return (all(cx["meta"][k] == cy["meta"][k] for k in keys) and cx["relu"] == cy["relu"] and cx["tile"] == cy["tile"]
                    and cx["splitk"] == cy["splitk"] and cx["math"] == cy["math"])

        i = first
        while i < len(ops):
            if not chainable(ops[i]):
                i += 1
                continue
            j = i
            while j + 1 < len(ops) and chainable(ops[j + 1]) and same_launch(ops[i], ops[j + 1]) and feeds(ops[j], ops[j + 1]):
                j += 1
            if j == i:
                i += 1
                continue
            run = ops[i:j + 1]
            segs, base = [], 0
            for k, op in enumerate(run):
                n = len(op.desc["segs"])
                for q, sg in enumerate(op.desc["segs"]):
                    segs.append(dict(sg, dep=(base - n + q) if k > 0 else None))
                base += n
            c = run[0].ctor
            names = [op.name for op in run]
            pre = os.path.commonprefix(names)
            name = names[0] + "".join("+" + nm[len(pre):] for nm in names[1:]) if pre else "+".join(names)
            merged = ConvOp(self, c["meta"], 1, 1, segs, c["relu"], tile=c["tile"], splitk=c["splitk"], name=name, math=c["math"], chain=True)
            merged.branch, merged.joins = run[0].branch, run[0].joins
            merged.parts = names  # the launches this one replaces (tools, profiles)
            ops[i:j + 1] = [merged]
            i += 1

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
        self._ensure_readback()
        self._rb_pending, self._rb_cache = False, None  # (a new forward: what an earlier fetch delivered is stale)
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
        if last is None:  # the forward's tail: pack what the host reads afterwards (captured with the rest)
            self._launch_readback(st)

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
        self.fetch()  # the read-back copy rides behind the forward: collect() finds counts / status in pinned memory

    @property
    def conv_macs(self):
        return sum(op.macs for op in self.ops)

    def describe(self):
        return [op.info for op in self.ops if isinstance(op, ConvOp)]

