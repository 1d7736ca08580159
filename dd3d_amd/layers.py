"""Parameter containers with the reference's state-dict layout.

These modules own the weights under exactly the names a reference checkpoint uses (SURVEY.md 8b
"Weights"), so ``load_state_dict`` of a tridet checkpoint works unchanged.  They deliberately have
NO torch ``forward``: all arithmetic of the forward path runs in the HIP library
(dd3d_amd/csrc) driven by ``dd3d_amd.engine``; there is no CPU / eager fallback.

Mirrors: detectron2.layers.Conv2d / FrozenBatchNorm2d / get_norm [ext], and
tridet/layers/normalization.py:12-40 (Scale, Offset, ModuleListDial).
"""
import math

import torch
from torch import nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the forward path runs in the HIP engine "
            "(dd3d_amd.engine) and has no eager fallback."
        )


class FrozenBatchNorm2d(_NoForward):
    """[ext] detectron2 FrozenBatchNorm2d: buffers weight/bias/running_mean/running_var, eps=1e-5."""
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)


class BatchNorm2d(_NoForward):
    """nn.BatchNorm2d layout (eval-mode use only): params weight/bias, buffers running_mean/var,
    num_batches_tracked."""
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(num_features), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


def get_norm(norm, out_channels):
    """[ext] detectron2.layers.get_norm for the values the reference configs use."""
    if norm is None or norm == "":
        return None
    if norm == "BN":
        return BatchNorm2d(out_channels)
    if norm == "FrozenBN":
        return FrozenBatchNorm2d(out_channels)
    raise ValueError(f"norm {norm!r} is not supported by the forward path")


class ModuleListDial(nn.ModuleList):
    """tridet/layers/normalization.py:30-40.  The reference's stateful round-robin (call k uses module
    k mod L) is realised statically here: the engine folds norm[l] into the epilogue of level l."""


class Conv2d(_NoForward):
    """[ext] detectron2.layers.Conv2d parameter layout: weight (O,I,kh,kw), optional bias, optional ``norm``."""
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, norm=None, groups=1):
        super().__init__()
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(
            torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size), requires_grad=False
        )
        self.bias = nn.Parameter(torch.zeros(out_channels), requires_grad=False) if bias else None
        if norm is not None:
            self.norm = norm
        else:
            self.norm = None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class Scale(_NoForward):
    """tridet/layers/normalization.py:12-18."""
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]), requires_grad=False)


class Offset(_NoForward):
    """tridet/layers/normalization.py:21-27."""
    def __init__(self, init_value=0.):
        super().__init__()
        self.bias = nn.Parameter(torch.FloatTensor([init_value]), requires_grad=False)


def fold_norm(conv, norm=None):
    """Per-output-channel (scale, shift) such that norm(conv_nobias(x) + bias) == conv_nobias(x)*scale + shift.
    [ext] FrozenBatchNorm2d / eval BatchNorm2d: y = (x-mean)*w*rsqrt(var+eps) + b."""
    n = conv.out_channels
    w = conv.weight
    scale = torch.ones(n, dtype=torch.float32, device=w.device)
    shift = torch.zeros(n, dtype=torch.float32, device=w.device)
    if conv.bias is not None:
        shift = conv.bias.detach().float().clone()
    norm = norm if norm is not None else conv.norm
    if norm is not None and not isinstance(norm, nn.ModuleList):
        s = norm.weight.detach().float() * torch.rsqrt(norm.running_var.float() + norm.eps)
        shift = (shift - norm.running_mean.float()) * s + norm.bias.detach().float()
        scale = s
    return scale, shift
