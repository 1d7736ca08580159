"""VoVNet-V2 (V-99-eSE and the other non-depthwise specs) + FPN(P2..P6) backbone: parameter tree and builders.

Module / parameter names follow tridet/modeling/feature_extractor/vovnet.py:79-97,124-161,188-335 exactly -- including
the ``<name>/conv`` / ``<name>/norm`` children of each nn.Sequential -- so reference checkpoints load unchanged
(``backbone.bottom_up.stage3.OSA3_2.layers.4.OSA3_2_4/conv.weight`` ...).  No torch forward: dd3d_amd.engine walks the
tree (``_vovnet``) and emits the HIP launches.
"""
from collections import OrderedDict

import torch
from torch import nn

from dd3d_amd.layers import Conv2d, get_norm
from dd3d_amd.modeling.dla import FPN, LastLevelP6
from dd3d_amd.registry import BACKBONE_REGISTRY
from dd3d_amd.structures import ShapeSpec

# vovnet.py:49-87 (non-depthwise specs; the dw / slim-dw variants are used by no reference config)
_STAGE_SPECS = {
    # depthwise variants (vovnet.py:18-36): stem_2 / stem_3 and every OSA layer are depthwise 3x3 -> pointwise 1x1 -> norm -> relu
    "V-19-slim-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512], layer_per_block=3,
                             block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024], layer_per_block=3,
                        block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512], layer_per_block=3,
                          block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024], layer_per_block=3,
                     block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024], layer_per_block=5,
                     block_per_stage=[1, 1, 2, 2]),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024], layer_per_block=5,
                     block_per_stage=[1, 1, 4, 3]),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024], layer_per_block=5,
                     block_per_stage=[1, 3, 9, 3]),
}


def _conv_norm(in_ch, out_ch, name, postfix, norm, kernel_size, stride=1):
    """vovnet.py:124-161 conv3x3 / conv1x1: [("<n>/conv", Conv2d no bias), ("<n>/norm", norm), ("<n>/relu", ReLU)]."""
    conv = Conv2d(in_ch, out_ch, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)
    nn.init.kaiming_normal_(conv.weight)  # vovnet.py:339-342
    seq = nn.Sequential(OrderedDict([(f"{name}_{postfix}/conv", conv), (f"{name}_{postfix}/norm", get_norm(norm, out_ch))]))
    seq.conv_name, seq.norm_name = f"{name}_{postfix}/conv", f"{name}_{postfix}/norm"
    return seq


def _dw_conv_norm(in_ch, out_ch, name, postfix, norm, stride=1):
    """vovnet.py:99-121 dw_conv3x3: [("<n>/dw_conv3x3", depthwise 3x3, groups = out_ch, no bias), ("<n>/pw_conv1x1", 1x1 no bias),
    ("<n>/pw_norm", norm), ("<n>/pw_relu", ReLU)]."""
    dw = Conv2d(in_ch, out_ch, 3, stride=stride, padding=1, bias=False, groups=out_ch)
    pw = Conv2d(in_ch, out_ch, 1, bias=False)
    nn.init.kaiming_normal_(dw.weight), nn.init.kaiming_normal_(pw.weight)
    n = f"{name}_{postfix}"
    seq = nn.Sequential(OrderedDict([(f"{n}/dw_conv3x3", dw), (f"{n}/pw_conv1x1", pw), (f"{n}/pw_norm", get_norm(norm, out_ch))]))
    seq.dw_name, seq.conv_name, seq.norm_name = f"{n}/dw_conv3x3", f"{n}/pw_conv1x1", f"{n}/pw_norm"
    return seq


def seq_dw(seq):
    """The depthwise 3x3 in front of the (pointwise) conv of a dw_conv3x3 sequence, or None."""
    return getattr(seq, seq.dw_name) if hasattr(seq, "dw_name") else None


def seq_conv(seq):
    return getattr(seq, seq.conv_name)


def seq_norm(seq):
    return getattr(seq, seq.norm_name)


class eSEModule(nn.Module):
    """vovnet.py:173-185: x * hsigmoid(fc(avg_pool(x))); fc = 1x1 conv with bias."""
    def __init__(self, channel):
        super().__init__()
        self.fc = Conv2d(channel, channel, 1, bias=True)
        nn.init.kaiming_normal_(self.fc.weight)


class OSAModule(nn.Module):
    """vovnet.py:188-238 (_OSA_module).  depthwise: a 1x1 `conv_reduction` to stage_ch first (iff in_ch != stage_ch), then dw_conv3x3 layers."""
    def __init__(self, in_ch, stage_ch, concat_ch, layer_per_block, module_name, norm, identity=False, depthwise=False):
        super().__init__()
        self.identity, self.in_ch, self.stage_ch, self.concat_ch, self.depthwise = identity, in_ch, stage_ch, concat_ch, depthwise
        self.layers = nn.ModuleList()
        c = in_ch
        self.conv_reduction = None
        if depthwise and in_ch != stage_ch:
            self.conv_reduction = _conv_norm(in_ch, stage_ch, f"{module_name}_reduction", "0", norm, 1)
        for i in range(layer_per_block):
            self.layers.append(_dw_conv_norm(stage_ch, stage_ch, module_name, i, norm) if depthwise else _conv_norm(c, stage_ch, module_name, i, norm, 3))
            c = stage_ch
        self.concat = _conv_norm(in_ch + layer_per_block * stage_ch, concat_ch, module_name, "concat", norm, 1)
        self.ese = eSEModule(concat_ch)


class OSAStage(nn.Sequential):
    """vovnet.py:241-273: MaxPool2d(3, 2, ceil_mode=True) before every stage but stage2, then the OSA modules."""
    def __init__(self, in_ch, stage_ch, concat_ch, block_per_stage, layer_per_block, stage_num, norm, depthwise=False):
        super().__init__()
        self.has_pool = stage_num != 2
        name = f"OSA{stage_num}_1"
        self.add_module(name, OSAModule(in_ch, stage_ch, concat_ch, layer_per_block, name, norm, depthwise=depthwise))
        for i in range(block_per_stage - 1):
            name = f"OSA{stage_num}_{i + 2}"
            self.add_module(name, OSAModule(concat_ch, stage_ch, concat_ch, layer_per_block, name, norm, identity=True, depthwise=depthwise))


class VoVNet(nn.Module):
    """vovnet.py:276-373."""
    def __init__(self, cfg, input_ch, out_features=None):
        super().__init__()
        if cfg.NAME not in _STAGE_SPECS:
            raise NotImplementedError(f"unknown VoVNet spec {cfg.NAME}; available: {sorted(_STAGE_SPECS)}")
        spec = _STAGE_SPECS[cfg.NAME]
        dw = bool(spec.get("dw", False))
        norm = cfg.NORM
        stem_ch = spec["stem"]
        self._out_features = list(out_features)
        stem = OrderedDict()
        self.stem_seqs = []
        for idx, (ci, co, st) in enumerate([(input_ch, stem_ch[0], 2), (stem_ch[0], stem_ch[1], 1), (stem_ch[1], stem_ch[2], 2)]):
            if dw and idx > 0:  # vovnet.py:301-305: stem_1 is a plain 3x3, stem_2 / stem_3 depthwise-separable
                s = _dw_conv_norm(ci, co, "stem", str(idx + 1), norm, stride=st)
                stem[s.dw_name] = seq_dw(s)
            else:
                s = _conv_norm(ci, co, "stem", str(idx + 1), norm, 3, stride=st)
            stem[s.conv_name], stem[s.norm_name] = seq_conv(s), seq_norm(s)
            self.stem_seqs.append((s.conv_name, s.norm_name, getattr(s, "dw_name", None)))
        self.stem = nn.Sequential(stem)
        self._out_feature_strides = {"stem": 4, "stage2": 4}
        self._out_feature_channels = {"stem": stem_ch[2]}
        in_ch_list = [stem_ch[2]] + spec["stage_out_ch"][:-1]
        self.stage_names = []
        stride = 4
        for i in range(4):
            name = f"stage{i + 2}"
            self.stage_names.append(name)
            self.add_module(
                name,
                OSAStage(in_ch_list[i], spec["stage_conv_ch"][i], spec["stage_out_ch"][i], spec["block_per_stage"][i],
                         spec["layer_per_block"], i + 2, norm, depthwise=dw)
            )
            self._out_feature_channels[name] = spec["stage_out_ch"][i]
            if i != 0:
                stride *= 2
                self._out_feature_strides[name] = stride

    def output_shape(self):
        return {
            n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features
        }

    @property
    def size_divisibility(self):
        return 0


@BACKBONE_REGISTRY.register()
def build_vovnet_backbone(cfg, input_shape):
    """vovnet.py:376-385 (cfg = cfg.FE.BACKBONE)."""
    return VoVNet(cfg, input_shape.channels, out_features=cfg.OUT_FEATURES)


@BACKBONE_REGISTRY.register()
def build_fcos_vovnet_fpn_backbone_p6(cfg, input_shape):
    """vovnet.py:428-454."""
    bottom_up = build_vovnet_backbone(cfg.FE.BACKBONE, input_shape)
    out_channels = cfg.FE.FPN.OUT_CHANNELS
    backbone = FPN(
        bottom_up=bottom_up, in_features=cfg.FE.FPN.IN_FEATURES, out_channels=out_channels, norm=cfg.FE.FPN.NORM,
        top_block=LastLevelP6(out_channels, out_channels, "p5"), fuse_type=cfg.FE.FPN.FUSE_TYPE
    )
    backbone._size_divisibility *= 2
    return backbone
