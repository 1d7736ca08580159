"""Model side of the drop-in boundary: importing this package registers the meta-architectures and backbone
builders under the names the reference registers (tridet/modeling/__init__.py, feature_extractor/__init__.py)."""
from dd3d_amd.modeling import dla, vovnet  # noqa: F401  (BACKBONE_REGISTRY: build_fcos_dla_fpn_backbone_p67, build_fcos_vovnet_fpn_backbone_p6, ...)
from dd3d_amd.modeling.dd3d import DD3D, build_feature_extractor  # noqa: F401  (META_ARCH_REGISTRY: DD3D)
from dd3d_amd.modeling.nuscenes_dd3d import NuscenesDD3D, get_group_idxs  # noqa: F401  (META_ARCH_REGISTRY: NuscenesDD3D)
from dd3d_amd.modeling.dense_depth import DD3DDenseDepth  # noqa: F401  (META_ARCH_REGISTRY: DD3DDenseDepth)
