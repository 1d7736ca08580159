"""Plugin registries of the forward path.

The reference registers its meta-architectures and backbone builders in detectron2's
``META_ARCH_REGISTRY`` / ``BACKBONE_REGISTRY`` (tridet/modeling/dd3d/core.py:18-19,
nuscenes_dd3d.py:299-300, feature_extractor/dla.py:536, vovnet.py:428) and instantiates them with
detectron2's ``build_model(cfg)`` (scripts/train.py:48).  When detectron2 is importable its own
registries are used, so ``DD3D`` etc. become visible to an unmodified caller; otherwise a local
registry with the same ``register()/get()`` surface stands in.
"""
import torch


class Registry:
    """Same surface as fvcore.common.registry.Registry: ``@REG.register()`` and ``REG.get(name)``."""
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:

            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class

            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


try:  # pragma: no cover - detectron2 is not installed in the build image
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY  # noqa: F401
except Exception:  # ModuleNotFoundError or a broken install
    META_ARCH_REGISTRY = Registry("META_ARCH")
    BACKBONE_REGISTRY = Registry("BACKBONE")


def build_model(cfg):
    """[ext] detectron2.modeling.build_model: ``META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)``
    then ``.to(cfg.MODEL.DEVICE)`` (caller: scripts/train.py:48)."""
    import dd3d_amd.modeling  # noqa: F401  (populates the registries)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
