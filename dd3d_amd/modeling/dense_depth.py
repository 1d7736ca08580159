"""``DD3DDenseDepth`` meta-architecture (tridet/modeling/dd3d/dense_depth.py:17-163): the depth pre-training network -- the
``box3d_tower`` of the FCOS3D head with one 1-channel ``dense_depth`` predictor PER pyramid level, followed by the aligned
bilinear upsampling of every level to the input resolution and the focal-length scaling.

The reference's ``forward`` only defines the training branch (it ends in ``raise NotImplementedError()`` in eval mode,
dense_depth.py:162-163); everything it computes before the losses is inference math, exposed here as
``predict_dense_depth(batched_inputs)``; ``forward`` keeps the reference's eval-mode behaviour.
"""
import torch
from torch import nn

from dd3d_amd.layers import Conv2d, Offset, Scale
from dd3d_amd.modeling.dd3d import build_feature_extractor
from dd3d_amd.modeling.heads import _init_predictor, _init_tower, _make_tower
from dd3d_amd.registry import META_ARCH_REGISTRY


class DD3DDenseDepthHead(nn.Module):
    """dense_depth.py:17-100."""
    def __init__(self, cfg, input_shape):
        super().__init__()
        c3 = cfg.DD3D.FCOS3D
        self.in_strides = [s.stride for s in input_shape]
        self.num_levels = len(input_shape)
        self.mean_depth_per_level = torch.FloatTensor(list(c3.MEAN_DEPTH_PER_LEVEL))  # plain attributes, as in the reference
        self.std_depth_per_level = torch.FloatTensor(list(c3.STD_DEPTH_PER_LEVEL))
        self.scale_depth_by_focal_lengths_factor = c3.SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR
        self.use_scale = c3.USE_SCALE
        self.depth_scale_init_factor = c3.DEPTH_SCALE_INIT_FACTOR
        in_channels = input_shape[0].channels
        if c3.USE_DEFORMABLE:
            raise ValueError("Not supported yet.")
        self.box3d_tower = _make_tower(in_channels, c3.NUM_CONVS, c3.NORM, self.num_levels)
        # each FPN level has its own predictor layer (dense_depth.py:63-67)
        self.dense_depth = nn.ModuleList([Conv2d(in_channels, 1, 3, 1, 1, bias=not self.use_scale) for _ in range(self.num_levels)])
        if self.use_scale:
            self.scales_depth = nn.ModuleList([
                Scale(init_value=float(sigma) * self.depth_scale_init_factor) for sigma in self.std_depth_per_level
            ])
            self.offsets_depth = nn.ModuleList([Offset(init_value=float(b)) for b in self.mean_depth_per_level])
        _init_tower(self.box3d_tower)
        for m in self.dense_depth:
            _init_predictor(m)


@META_ARCH_REGISTRY.register()
class DD3DDenseDepth(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.in_features = cfg.DD3D.IN_FEATURES
        self.feature_locations_offset = cfg.DD3D.FEATURE_LOCATIONS_OFFSET
        self.backbone = build_feature_extractor(cfg)
        shapes = self.backbone.output_shape()
        self.backbone_output_shape = [shapes[f] for f in self.in_features]
        if list(self.in_features) != list(shapes.keys()):
            raise NotImplementedError("DD3D.IN_FEATURES must select every FPN output (all reference configs do)")
        self.fcos3d_head = DD3DDenseDepthHead(cfg, self.backbone_output_shape)
        self.scale_depth_by_focal_lengths = cfg.DD3D.FCOS3D.SCALE_DEPTH_BY_FOCAL_LENGTHS
        self.scale_depth_by_focal_lengths_factor = cfg.DD3D.FCOS3D.SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR
        self.register_buffer("pixel_mean", torch.Tensor(list(cfg.MODEL.PIXEL_MEAN)).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(list(cfg.MODEL.PIXEL_STD)).view(-1, 1, 1))
        self._plans = {}
        self.use_graph = True
        self.math = None
        self.training = False

    @property
    def device(self):
        return self.pixel_mean.device

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("dd3d_amd implements the inference math only")
        return super().train(False)

    def invalidate_plans(self):
        """Plans read packed copies of the weights (one store per model, shared by its plans): drop both."""
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

    def get_plan(self, B, Hp, Wp):
        from dd3d_amd.engine import DenseDepthPlan
        key = (B, Hp, Wp, self.math)
        plan = self._plans.pop(key, None)
        if plan is None:
            plan = DenseDepthPlan(self, B, Hp, Wp)
            if self.use_graph:
                plan.capture()
        self._plans[key] = plan  # most recently used last; bounded like DD3D.get_plan
        while len(self._plans) > 8:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self._plans.pop(next(iter(self._plans)))
        return plan

    @torch.no_grad()
    def predict_dense_depth(self, batched_inputs):
        """dense_depth.py:121-151 up to (not including) the losses: a list over pyramid levels of (B, Hp, Wp) depth maps at the
        padded input resolution, on the model's device.  The default f16x2 arithmetic is guarded exactly as in DD3D.forward: a kernel
        that met an activation outside the half range trips the plan's status word, the maps are NOT returned, and a model on the
        default arithmetic re-runs (from then on) with the three-term bf16 split."""
        from dd3d_amd.engine import relax_arithmetic
        while True:
            try:
                return self._predict_dense_depth(batched_inputs)
            except FloatingPointError as e:
                if not relax_arithmetic(self, e):  # (plane scale 16 -> 4 -> 1, then bf16x3; an explicitly chosen arithmetic raises)
                    raise

    def _predict_dense_depth(self, batched_inputs):
        images = [x["image"] for x in batched_inputs]
        div = self.backbone.size_divisibility
        H = max(int(im.shape[-2]) for im in images)
        W = max(int(im.shape[-1]) for im in images)
        if div > 1:
            H, W = (H + div - 1) // div * div, (W + div - 1) // div * div
        B = len(images)
        plan = self.get_plan(B, H, W)
        for i, im in enumerate(images):
            assert im.dtype == torch.uint8 and im.shape[0] == 3
            plan.in_u8[i, :, :im.shape[1], :im.shape[2]].copy_(im, non_blocking=True)
        plan.in_sizes.copy_(torch.tensor([[int(im.shape[-2]), int(im.shape[-1])] for im in images], dtype=torch.int32), non_blocking=True)
        if self.scale_depth_by_focal_lengths:
            if "intrinsics" not in batched_inputs[0]:
                raise AssertionError("SCALE_DEPTH_BY_FOCAL_LENGTHS needs 'intrinsics'")  # dense_depth.py:147
            K = torch.stack([x["intrinsics"].float().cpu() for x in batched_inputs], 0)
            if torch.allclose(K[0], torch.eye(3)):
                raise ValueError("Intrinsics is Identity.")  # image_list.py:57-62
            plan.in_K.copy_(K.reshape(B, 9), non_blocking=True)
        plan.run()
        plan.check_status()  # one 4-byte read behind the forward (the caller is about to consume the maps anyway); raises and clears
        return [m for m in plan.depth_maps]

    def forward(self, batched_inputs):
        raise NotImplementedError()  # the reference's eval-mode forward (dense_depth.py:162-163)
