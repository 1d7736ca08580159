"""``DD3D`` meta-architecture: the drop-in boundary of the forward path.

Same registered name, constructor signature ``(cfg)``, attributes and call contract as
tridet/modeling/dd3d/core.py:18-175:  ``model(batched_inputs: List[dict]) -> List[{"instances": Instances}]``.
Only the inference branch exists here (training is out of scope); everything from the
uint8 image to the final detections runs in the HIP engine (dd3d_amd.engine) on the model's device.
"""
import numpy as np
import torch
from torch import nn

from dd3d_amd.engine import ForwardPlan
from dd3d_amd.modeling.heads import FCOS2DHead, FCOS3DHead
from dd3d_amd.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY
from dd3d_amd.structures import Boxes, Boxes3D, Instances, ShapeSpec


_EYE3 = np.eye(3, dtype=np.float32)


def _as_f32_array(t):
    """Host float32 numpy view / copy of a small tensor-like (the per-image intrinsics)."""
    if isinstance(t, torch.Tensor):
        return t.detach().to(device="cpu", dtype=torch.float32).numpy()
    return np.asarray(t, dtype=np.float32)


def build_feature_extractor(cfg, input_shape=None):
    """tridet/modeling/feature_extractor/__init__.py:13-26."""
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.FE.BUILDER)(cfg, input_shape)


@META_ARCH_REGISTRY.register()
class DD3D(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone = build_feature_extractor(cfg)
        backbone_output_shape = self.backbone.output_shape()
        self.in_features = cfg.DD3D.IN_FEATURES or list(backbone_output_shape.keys())
        self.backbone_output_shape = [backbone_output_shape[f] for f in self.in_features]
        self.feature_locations_offset = cfg.DD3D.FEATURE_LOCATIONS_OFFSET

        self.fcos2d_head = FCOS2DHead(cfg, self.backbone_output_shape)
        if cfg.MODEL.BOX3D_ON:
            self.fcos3d_head = FCOS3DHead(cfg, self.backbone_output_shape)
            self.only_box2d = False
        else:
            self.only_box2d = True

        self.postprocess_in_inference = cfg.DD3D.INFERENCE.DO_POSTPROCESS
        self.do_nms = cfg.DD3D.INFERENCE.DO_NMS
        self.do_bev_nms = cfg.DD3D.INFERENCE.DO_BEV_NMS
        self.bev_nms_iou_thresh = cfg.DD3D.INFERENCE.BEV_NMS_IOU_THRESH
        self.nusc_sample_aggregate_in_inference = cfg.DD3D.INFERENCE.NUSC_SAMPLE_AGGREGATE
        self.num_classes = cfg.DD3D.NUM_CLASSES

        self.register_buffer("pixel_mean", torch.Tensor(list(cfg.MODEL.PIXEL_MEAN)).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(list(cfg.MODEL.PIXEL_STD)).view(-1, 1, 1))
        self._plans = {}
        self.max_cached_plans = 8  # launch plans kept per model (one per (batch, padded size, flags)); least recently used goes first
        self.use_graph = True
        # None: DD3D_MATH / the default ("f16x2", falling back to "bf16x3" for good if an activation ever leaves the half range); or one
        # of "f16x2" / "bf16x3" / "f32" / "bf16x2" / "bf16" (dd3d_amd.engine.default_math), set before the first forward
        self.math = None
        # plane scale of the f16x2 arithmetic (a power of two; None: DD3D_F16_ACT_SCALE / 16).  The range guard's staged fallback lowers it
        # 16 -> 4 -> 1 on an overflow before it gives up the f16x2 speed (dd3d_amd.engine.plan.relax_arithmetic)
        self.act_scale = None
        # None / "latency": launch plans tiled for one forward at a time; "throughput": for plans that share the chip with other plans in
        # flight (dd3d_amd.engine.tiling.THROUGHPUT_TILE_TABLE; PipelinedForward slots covering several requests choose it themselves)
        self.tile_policy = None
        self.training = False

    @property
    def device(self):
        return self.pixel_mean.device

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("dd3d_amd implements the inference forward path only")
        return super().train(False)

    # ------------------------------------------------------------------ plan management
    def invalidate_plans(self):
        """Call after changing weights in place (plans read packed copies of the weights, shared by all plans of the model)."""
        self._plans = {}
        self.__dict__.pop("_weight_store", None)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_plans()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate_plans()
        return r

    def _sync_flags(self):
        """`do_test` / TTA flip these attributes on the instance (scripts/train.py:206-209,268-272); mirror them
        into the config the plan is built from."""
        inf = self.cfg.DD3D.INFERENCE
        inf.DO_POSTPROCESS, inf.DO_NMS = bool(self.postprocess_in_inference), bool(self.do_nms)
        inf.DO_BEV_NMS, inf.BEV_NMS_IOU_THRESH = bool(self.do_bev_nms), float(self.bev_nms_iou_thresh)
        return (bool(self.postprocess_in_inference), bool(self.do_nms), bool(self.do_bev_nms))

    def get_plan(self, B, Hp, Wp, world_size=1, rank=0, exchange=None, camera_sharded=False):
        exchange = world_size > 1 if exchange is None else bool(exchange)
        key = (B, Hp, Wp, world_size, rank, exchange, bool(camera_sharded), self.math, self.act_scale, getattr(self, "tile_policy", None)) + self._sync_flags()
        plan = self._plans.pop(key, None)
        if plan is None:
            plan = ForwardPlan(self, B, Hp, Wp, world_size=world_size, rank=rank, exchange=exchange, camera_sharded=camera_sharded)
            if self.use_graph and not exchange:
                plan.capture()
        self._plans[key] = plan  # (re-)inserted last = most recently used
        self._evict_plans()
        return plan

    def _evict_plans(self):
        """Inputs of ever-changing size would otherwise pin a buffer set (0.5 - 3 GB) per size for the life of the model."""
        while len(self._plans) > max(1, int(self.max_cached_plans)):
            if torch.cuda.is_available():
                torch.cuda.synchronize()  # its last replay may still be running on the buffers about to be released
            self._plans.pop(next(iter(self._plans)))

    # ------------------------------------------------------------------ host side of forward
    def stage_inputs(self, batched_inputs, plan=None, first=0, partial=False, flush=True):
        """core.py:65-72: gather images/intrinsics; the padded canvas geometry is ImageList.from_tensors'
        (image_list.py:120-142).  Returns (plan, image_sizes).  `first` (with a fixed `plan` only): the batch goes to positions
        [first, first + len(batch)) of a plan; `partial` allows a batch shorter than the plan's (PipelinedForward micro-batches, the
        short last batch of a shard): the other positions keep what they held and their outputs are the caller's to ignore.

        Host cost per request (round 6): the image goes by ONE asynchronous copy per image (a pinned source never blocks the host); sizes,
        intrinsics, resize targets (and poses / sample ids of a model with BEV stages) are plain stores into the plan's pinned host mirrors,
        shipped by `plan.flush_inputs()` -- here when `flush`, or once per slot run by a runner that stages several requests into one plan
        (`flush=False`; PipelinedForward._enqueue)."""
        images = [x["image"] for x in batched_inputs]
        image_sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        div = self.backbone.size_divisibility
        H = max(s[0] for s in image_sizes)
        W = max(s[1] for s in image_sizes)
        if div > 1:
            H, W = (H + div - 1) // div * div, (W + div - 1) // div * div
        B = len(images)
        if "intrinsics" not in batched_inputs[0]:
            if not self.only_box2d:
                raise ValueError("DD3D with BOX3D_ON needs 'intrinsics' in every input dict")
            K = np.tile(np.eye(3, dtype=np.float32) * 2.0, (B, 1, 1))
        else:
            K = np.stack([_as_f32_array(x["intrinsics"]) for x in batched_inputs], 0)
            if np.allclose(K[0], _EYE3, rtol=1e-5, atol=1e-8):
                raise ValueError("Intrinsics is Identity.")  # image_list.py:57-62
        if plan is None:
            assert first == 0 and not partial
            plan = self.get_plan(B, H, W)
        elif (H, W) != (plan.Hp, plan.Wp) or first < 0 or first + B > plan.B or (B != plan.B and not partial):
            # a fixed plan (DistributedForward / PipelinedForward): a short batch would leave stale images in the other positions (allowed
            # only where the caller says so: `partial`), a smaller canvas would run on a larger padded canvas than the reference's
            # ImageList (other border features)
            raise ValueError(f"batch of {B} images on a {H}x{W} canvas at position {first} does not fit the fixed launch plan "
                             f"(B={plan.B}, {plan.Hp}x{plan.Wp}); pad the batch / canvas or build a plan for this geometry")
        sl = slice(first, first + B)
        for i, im in enumerate(images):
            assert im.dtype == torch.uint8 and im.shape[0] == 3, "expected uint8 (3,H,W) images (dataset_mapper.py:127)"
            if image_sizes[i] == (plan.Hp, plan.Wp):
                plan.in_u8[first + i].copy_(im, non_blocking=True)  # whole canvas: one contiguous copy
            else:
                plan.in_u8[first + i, :, :im.shape[1], :im.shape[2]].copy_(im, non_blocking=True)
        plan.inputs_writable()  # (the previous forward's copy out of the host mirrors has run: normally long ago)
        m = plan.host_np  # numpy views of the pinned mirrors: plain host stores, no tensor is built per request
        m["sizes"][sl] = image_sizes
        m["K"][sl] = K.reshape(B, 9)
        m["outsize"][sl] = [(s[0], s[1], x.get("height", s[0]), x.get("width", s[1])) for s, x in zip(image_sizes, batched_inputs)]
        if plan.has_bev_inputs:  # BEV stages need camera->global poses and sample membership
            m["pose"][sl] = [self._pose_vec(x) for x in batched_inputs]
            if not getattr(plan, "camera_sharded", False):  # (camera-sharded: membership is positional in the global image order)
                # sample ids are per request: offset by the position so that requests sharing a plan never merge their samples
                m["group"][sl] = np.asarray(self._sample_groups(batched_inputs), dtype=np.int32) + first
        if flush:
            plan.flush_inputs()
        return plan, image_sizes

    @staticmethod
    def _pose_vec(x):
        """core.py:138-141: 'pose' if present, else 'extrinsics'; (quat wxyz, tvec) as postprocessing.py:34 reads them."""
        pose = x["pose"] if "pose" in x else x["extrinsics"]
        return [float(v) for v in pose.quat.elements] + [float(v) for v in pose.tvec]

    def _sample_groups(self, batched_inputs):
        return list(range(len(batched_inputs)))

    def _counts(self, plan):
        """Detection counts of the plan's last forward (this is the host's wait for the forward), with the device-side faults checked --
        all out of the forward's read-back record (engine.PlanBase.readback: one asynchronous copy into pinned memory, enqueued behind the
        forward; round 5 read counts, status word and range-guard maxima with one blocking copy each, per request)."""
        rb = plan.readback()
        if not getattr(rb, "checked", False):  # (once per forward: the requests sharing a slot run read the same record)
            plan.check_status(rb)  # a numeric fault flagged by a kernel (half-range overflow of the f16x2 mode) fails the forward loudly
            rb.checked = True
        counts = rb.counts
        if counts.numel() and int(counts.min()) < 0:
            raise RuntimeError("more than 8192 detections met in one BEV NMS problem (the capacity of its LDS sorter): feed fewer images per step")
        n_max = int(counts.max()) if counts.numel() else 0
        if n_max > plan.det_cap:
            raise RuntimeError(f"{n_max} detections exceed the detection buffer ({plan.det_cap}); raise det_cap")
        return counts, n_max

    def _instances(self, plan, d, size, inv_K):
        """Rows of the detection buffer (a PRIVATE copy: the fields below are views of it, no further copies) -> Instances with the
        reference's fields (core.py:153-164)."""
        n = d.shape[0]
        r = Instances(size)
        r.pred_boxes = Boxes(d[:, 0:4])
        r.scores = d[:, 4]
        ints = d[:, 6:8].to(torch.int64)  # [class, level]
        r.pred_classes = ints[:, 0]
        r.locations = d[:, 8:10]
        r.fpn_levels = ints[:, 1]
        if not self.only_box2d:
            r.pred_boxes3d = Boxes3D(d[:, 10:14], d[:, 14:16], d[:, 16:17], d[:, 17:20], inv_K[None].expand(n, 3, 3))
            r.scores_3d = d[:, 5]
        self._collect_extra(r, d, plan)
        return r

    def collect(self, plan, batched_inputs, image_sizes, first=0):
        """Detection buffer -> List[{"instances": Instances}] for the images at positions [first, first + len(batch)) of the plan.  Per image:
        one device copy of its detection rows (the results must not alias a buffer the next forward overwrites) and one integer cast;
        every field is a view of that copy."""
        counts, n_max = self._counts(plan)
        B = len(batched_inputs)
        inv_K = plan.inv_K.view(-1, 3, 3)[first:first + B].clone()
        results = []
        for i, (inp, isz) in enumerate(zip(batched_inputs, image_sizes)):
            g = first + i
            d = plan.det[g, :int(counts[g])].clone()
            size = (int(inp.get("height", isz[0])), int(inp.get("width", isz[1]))) if self.postprocess_in_inference else isz
            results.append({"instances": self._instances(plan, d, size, inv_K[i])})
        return results

    def collect_owned(self, plan):
        """Camera-sharded plans (engine.ForwardPlan(camera_sharded=True)): the detections of the samples THIS rank owns, as
        [(global image index, {"instances": Instances})] -- cameras that other ranks decoded included.  Everything comes from the
        device: counts and detections from the owner's post stages, image sizes and K^-1 from the gathered records."""
        assert plan.camera_sharded
        if plan.G == 0:
            torch.cuda.current_stream().synchronize()
            return []
        counts, n_max = self._counts(plan)
        det = plan.det
        sl = slice(plan.img_first, plan.img_first + plan.G)
        osz = plan.gathered_field("outsize")[sl].cpu()
        inv_K = plan.gathered_field("inv_K")[sl].reshape(-1, 3, 3).clone()
        results = []
        for g in range(plan.G):
            d = det[g, :int(counts[g])].clone()
            size = (int(osz[g, 2]), int(osz[g, 3])) if self.postprocess_in_inference else (int(osz[g, 0]), int(osz[g, 1]))
            results.append((plan.img_first + g, {"instances": self._instances(plan, d, size, inv_K[g])}))
        return results

    def _collect_extra(self, r, d, plan):
        pass

    @torch.no_grad()
    def forward(self, batched_inputs):
        from dd3d_amd.engine import relax_arithmetic
        while True:
            plan, image_sizes = self.stage_inputs(batched_inputs)
            plan.run()
            try:
                return self.collect(plan, batched_inputs, image_sizes)
            except FloatingPointError as e:
                # The range guard of the f16x2 arithmetic.  A model on the default arithmetic first widens the half range (plane scale 16 -> 4
                # -> 1: activations up to 4094 -> 16376 -> 65504 at the same speed), then runs on the three-term bf16 split from now on (same
                # f32-equivalent products, full f32 exponent range, twice the matrix work); an explicitly chosen arithmetic raises.
                if not relax_arithmetic(self, e):
                    raise
