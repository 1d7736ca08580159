"""dd3d_amd -- MI355X-native DD3D inference forward path (drop-in for tridet.modeling.dd3d on that path).

    from dd3d_amd import get_cfg, build_model
    model = build_model(get_cfg("dd3d_kitti_dla34"))      # META_ARCH_REGISTRY["DD3D"](cfg).to("cuda")
    outputs = model(batched_inputs)                        # [{"instances": Instances}, ...]

All arithmetic runs in the HIP library built from dd3d_amd/csrc (C ABI: include/dd3d_hip.h).
"""
from dd3d_amd.config import CfgNode, get_cfg  # noqa: F401
from dd3d_amd.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, build_model  # noqa: F401

__version__ = "0.1.0"
