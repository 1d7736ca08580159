"""FCOS2D / FCOS3D head parameter trees.

Naming and constructor logic follow tridet/modeling/dd3d/fcos2d.py:30-128 and fcos3d.py:55-158 so that
reference checkpoints load unchanged; no torch forward (see dd3d_amd.engine for the HIP plan).
"""
import torch
from torch import nn

from dd3d_amd.layers import Conv2d, ModuleListDial, Offset, Scale, get_norm


def _init_tower(tower):
    for l in tower.modules():
        if isinstance(l, Conv2d):
            nn.init.kaiming_normal_(l.weight, mode="fan_out", nonlinearity="relu")
            if l.bias is not None:
                nn.init.constant_(l.bias, 0)


def _init_predictor(conv):
    nn.init.kaiming_uniform_(conv.weight, a=1)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


def _make_tower(in_channels, num_convs, norm, num_levels):
    """fcos2d.py:71-93 (v2) / fcos3d.py:81-101: conv3x3 (bias iff no norm) + per-level norm + relu."""
    tower = []
    for _ in range(num_convs):
        if norm in ("BN", "FrozenBN"):
            norm_layer = ModuleListDial([get_norm(norm, in_channels) for _ in range(num_levels)])
        else:
            norm_layer = get_norm(norm, in_channels)
        tower.append(Conv2d(in_channels, in_channels, 3, 1, 1, bias=norm_layer is None, norm=norm_layer))
    return nn.Sequential(*tower)


class FCOS2DHead(nn.Module):
    """fcos2d.py:30-128."""
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.num_classes = cfg.DD3D.NUM_CLASSES
        self.in_strides = [s.stride for s in input_shape]
        self.num_levels = len(input_shape)
        self.use_scale = cfg.DD3D.FCOS2D.USE_SCALE
        self._version = cfg.DD3D.FCOS2D._VERSION
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1, "Each level must have the same channel!"
        in_channels = in_channels[0]
        if cfg.DD3D.FCOS2D.USE_DEFORMABLE:
            raise ValueError("Not supported yet.")
        if self._version == "v1":  # fcos2d.py:58-70: legacy tower layout (conv with bias, separate norm / ReLU modules)
            raise NotImplementedError("FCOS2D _VERSION v1 is not built (no reference config uses it; configs/models/dd3d.yaml sets v2)")
        if self._version != "v2":
            raise ValueError(f"Invalid FCOS2D version: {self._version}")
        norm = cfg.DD3D.FCOS2D.NORM
        self.cls_tower = _make_tower(in_channels, cfg.DD3D.FCOS2D.NUM_CLS_CONVS, norm, self.num_levels)
        self.box2d_tower = _make_tower(in_channels, cfg.DD3D.FCOS2D.NUM_BOX_CONVS, norm, self.num_levels)
        self.cls_logits = Conv2d(in_channels, self.num_classes, 3, 1, 1)
        self.box2d_reg = Conv2d(in_channels, 4, 3, 1, 1)
        self.centerness = Conv2d(in_channels, 1, 3, 1, 1)
        if self.use_scale:
            f = cfg.DD3D.FCOS2D.BOX2D_SCALE_INIT_FACTOR
            self.scales_box2d_reg = nn.ModuleList([Scale(init_value=s * f) for s in self.in_strides])
        _init_tower(self.cls_tower), _init_tower(self.box2d_tower)
        for m in (self.cls_logits, self.box2d_reg, self.centerness):
            _init_predictor(m)


class FCOS3DHead(nn.Module):
    """fcos3d.py:55-158."""
    def __init__(self, cfg, input_shape):
        super().__init__()
        c3 = cfg.DD3D.FCOS3D
        self.num_classes = cfg.DD3D.NUM_CLASSES
        self.in_strides = [s.stride for s in input_shape]
        self.num_levels = len(input_shape)
        self.use_scale = c3.USE_SCALE
        self.use_per_level_predictors = c3.PER_LEVEL_PREDICTORS
        self.register_buffer("mean_depth_per_level", torch.Tensor(list(c3.MEAN_DEPTH_PER_LEVEL)))
        self.register_buffer("std_depth_per_level", torch.Tensor(list(c3.STD_DEPTH_PER_LEVEL)))
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1, "Each level must have the same channel!"
        in_channels = in_channels[0]
        if c3.USE_DEFORMABLE:
            raise ValueError("Not supported yet.")
        self.box3d_tower = _make_tower(in_channels, c3.NUM_CONVS, c3.NORM, self.num_levels)
        nc = self.num_classes if not c3.CLASS_AGNOSTIC_BOX3D else 1
        nl = self.num_levels if c3.PER_LEVEL_PREDICTORS else 1
        self.class_agnostic = bool(c3.CLASS_AGNOSTIC_BOX3D)

        def pred(k, bias=True):
            return nn.ModuleList([Conv2d(in_channels, k * nc, 3, 1, 1, bias=bias) for _ in range(nl)])

        self.box3d_quat, self.box3d_ctr = pred(4), pred(2)
        self.box3d_depth = pred(1, bias=not self.use_scale)  # fcos3d.py:116
        self.box3d_size, self.box3d_conf = pred(3), pred(1)
        if self.use_scale:
            self.scales_proj_ctr = nn.ModuleList([
                Scale(init_value=s * c3.PROJ_CTR_SCALE_INIT_FACTOR) for s in self.in_strides
            ])
            self.scales_size = nn.ModuleList([Scale(1.0) for _ in range(self.num_levels)])
            self.scales_conf = nn.ModuleList([Scale(1.0) for _ in range(self.num_levels)])
            self.scales_depth = nn.ModuleList([
                Scale(init_value=float(sigma) * c3.DEPTH_SCALE_INIT_FACTOR) for sigma in self.std_depth_per_level
            ])
            self.offsets_depth = nn.ModuleList([Offset(init_value=float(b)) for b in self.mean_depth_per_level])
        _init_tower(self.box3d_tower)
        for ml in (self.box3d_quat, self.box3d_ctr, self.box3d_depth, self.box3d_size, self.box3d_conf):
            for m in ml:
                _init_predictor(m)
