"""CPU ORACLE for the VoVNet-V2 bottom-up (test infrastructure only; see oracle/dd3d_oracle.py for the rules).

Restates tridet/modeling/feature_extractor/vovnet.py:99-367 (all seven specs) as functions over a state_dict."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle.dd3d_oracle import batch_norm_eval

DW_SPECS = {"V-19-slim-dw-eSE", "V-19-dw-eSE"}  # vovnet.py:18-36: stem [64, 64, 64], depthwise-separable layers
SPECS = {  # vovnet.py:18-87
    "V-19-slim-dw-eSE": ([64, 64, 64], [64, 80, 96, 112], [112, 256, 384, 512], 3, [1, 1, 1, 1]),
    "V-19-dw-eSE": ([64, 64, 64], [128, 160, 192, 224], [256, 512, 768, 1024], 3, [1, 1, 1, 1]),
    "V-19-slim-eSE": ([64, 64, 128], [64, 80, 96, 112], [112, 256, 384, 512], 3, [1, 1, 1, 1]),
    "V-19-eSE": ([64, 64, 128], [128, 160, 192, 224], [256, 512, 768, 1024], 3, [1, 1, 1, 1]),
    "V-39-eSE": ([64, 64, 128], [128, 160, 192, 224], [256, 512, 768, 1024], 5, [1, 1, 2, 2]),
    "V-57-eSE": ([64, 64, 128], [128, 160, 192, 224], [256, 512, 768, 1024], 5, [1, 1, 4, 3]),
    "V-99-eSE": ([64, 64, 128], [128, 160, 192, 224], [256, 512, 768, 1024], 5, [1, 3, 9, 3]),
}


def _cnr(sd, p, x, stride=1, padding=1, hook=None):
    """vovnet.py:124-161 conv3x3 / conv1x1: conv (no bias) -> norm -> ReLU, children named '<p>/conv', '<p>/norm'."""
    y = F.conv2d(x, sd[p + "/conv.weight"], None, stride=stride, padding=padding)
    return F.relu(batch_norm_eval(sd, p + "/norm", y, hook))


def _dw_cnr(sd, p, x, stride=1, hook=None):
    """vovnet.py:99-121 dw_conv3x3: depthwise 3x3 (groups = channels, no norm / relu) -> pointwise 1x1 -> norm -> ReLU."""
    w = sd[p + "/dw_conv3x3.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=1, groups=w.shape[0])
    y = F.conv2d(y, sd[p + "/pw_conv1x1.weight"], None)
    return F.relu(batch_norm_eval(sd, p + "/pw_norm", y, hook))


def _osa(sd, p, name, x, layers, identity, hook=None, depthwise=False):
    """vovnet.py:218-238 _OSA_module.forward + eSEModule.forward (vovnet.py:180-185) + Hsigmoid (vovnet.py:164-170)."""
    identity_feat = x
    outs = [x]
    if depthwise and f"{p}.conv_reduction.{name}_reduction_0/conv.weight" in sd:  # isReduced (vovnet.py:201-205,224-225)
        x = _cnr(sd, f"{p}.conv_reduction.{name}_reduction_0", x, padding=0, hook=hook)
    for i in range(layers):
        x = _dw_cnr(sd, f"{p}.layers.{i}.{name}_{i}", x, hook=hook) if depthwise else _cnr(sd, f"{p}.layers.{i}.{name}_{i}", x, hook=hook)
        outs.append(x)
    xt = _cnr(sd, f"{p}.concat.{name}_concat", torch.cat(outs, 1), padding=0, hook=hook)
    g = F.adaptive_avg_pool2d(xt, 1)
    g = F.conv2d(g, sd[p + ".ese.fc.weight"], sd[p + ".ese.fc.bias"])
    xt = xt * (F.relu6(g + 3.0) / 6.0)
    return xt + identity_feat if identity else xt


def vovnet_forward(sd, x, name, out_features, prefix="backbone.bottom_up", hook=None):
    """vovnet.py:357-367 VoVNet.forward."""
    stem_ch, _, _, layers, blocks = SPECS[name]
    p = prefix
    dw = name in DW_SPECS
    x = _cnr(sd, p + ".stem.stem_1", x, stride=2, hook=hook)
    x = _dw_cnr(sd, p + ".stem.stem_2", x, hook=hook) if dw else _cnr(sd, p + ".stem.stem_2", x, hook=hook)
    x = _dw_cnr(sd, p + ".stem.stem_3", x, stride=2, hook=hook) if dw else _cnr(sd, p + ".stem.stem_3", x, stride=2, hook=hook)
    outs = OrderedDict()
    for si in range(4):
        stage = si + 2
        if stage != 2:
            x = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)  # vovnet.py:248-249
        for b in range(blocks[si]):
            nm = f"OSA{stage}_{b + 1}"
            x = _osa(sd, f"{p}.stage{stage}.{nm}", nm, x, layers, identity=b > 0, hook=hook, depthwise=dw)
        if f"stage{stage}" in out_features:
            outs[f"stage{stage}"] = x
    return outs
