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
    """vovnet.py:188-238 (_OSA_module, depthwise False)."""
    def __init__(self, in_ch, stage_ch, concat_ch, layer_per_block, module_name, norm, identity=False):
        super().__init__()
        self.identity, self.in_ch, self.stage_ch, self.concat_ch = identity, in_ch, stage_ch, concat_ch
        self.layers = nn.ModuleList()
        c = in_ch
        for i in range(layer_per_block):
            self.layers.append(_conv_norm(c, stage_ch, module_name, i, norm, 3))
            c = stage_ch
        self.concat = _conv_norm(in_ch + layer_per_block * stage_ch, concat_ch, module_name, "concat", norm, 1)
        self.ese = eSEModule(concat_ch)


class OSAStage(nn.Sequential):
    """vovnet.py:241-273: MaxPool2d(3, 2, ceil_mode=True) before every stage but stage2, then the OSA modules."""
    def __init__(self, in_ch, stage_ch, concat_ch, block_per_stage, layer_per_block, stage_num, norm):
        super().__init__()
        self.has_pool = stage_num != 2
        name = f"OSA{stage_num}_1"
        self.add_module(name, OSAModule(in_ch, stage_ch, concat_ch, layer_per_block, name, norm))
        for i in range(block_per_stage - 1):
            name = f"OSA{stage_num}_{i + 2}"
            self.add_module(name, OSAModule(concat_ch, stage_ch, concat_ch, layer_per_block, name, norm, identity=True))


class VoVNet(nn.Module):
    """vovnet.py:276-373."""
    def __init__(self, cfg, input_ch, out_features=None):
        super().__init__()
        if cfg.NAME not in _STAGE_SPECS:
            raise NotImplementedError(f"VoVNet spec {cfg.NAME} (depthwise variants) is not used by any reference config")
        spec = _STAGE_SPECS[cfg.NAME]
        if any(c % 32 for c in spec["stage_conv_ch"] + spec["stage_out_ch"]):
            # the implicit-GEMM kernels address channel slices of the OSA concat buffers in 32-channel K chunks
            raise NotImplementedError(f"VoVNet spec {cfg.NAME}: channel counts that are not multiples of 32 are not built "
                                      "(V-19-eSE, V-39-eSE, V-57-eSE and V-99-eSE are; the reference's configs use V-99-eSE)")
        norm = cfg.NORM
        stem_ch = spec["stem"]
        self._out_features = list(out_features)
        stem = OrderedDict()
        self.stem_seqs = []
        for idx, (ci, co, st) in enumerate([(input_ch, stem_ch[0], 2), (stem_ch[0], stem_ch[1], 1), (stem_ch[1], stem_ch[2], 2)]):
            s = _conv_norm(ci, co, "stem", str(idx + 1), norm, 3, stride=st)
            stem[s.conv_name], stem[s.norm_name] = seq_conv(s), seq_norm(s)
            self.stem_seqs.append((s.conv_name, s.norm_name))
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
                         spec["layer_per_block"], i + 2, norm)
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
