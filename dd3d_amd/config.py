"""Config surface of the DD3D forward path.

The reference composes its config with Hydra/OmegaConf (configs/defaults.yaml:1-11); neither is
installed here, and only the keys the *forward path* reads matter for a drop-in
(SURVEY.md section 8b "Config surface").  ``CfgNode`` is an attribute-style mapping that accepts the
same key paths (``cfg.DD3D.FCOS2D.INFERENCE.NMS_THRESH`` ...), so an OmegaConf ``DictConfig`` from the
reference works as well (anything with attribute access does).

Values mirror configs/models/dd3d.yaml, configs/meta_arch/dd3d.yaml, configs/feature_extractors/*.yaml,
configs/train_datasets/{kitti_3d,nuscenes}.yaml and the experiment deltas
configs/experiments/dd3d_{kitti,nusc}_{dla34,v99}.yaml of the reference.
"""
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access, recursively."""
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else copy.deepcopy(v)
        return self

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))


_CONFIG_DIR = os.path.join(os.path.dirname(__file__), "configs")


def _load_yaml(name):
    with open(os.path.join(_CONFIG_DIR, name)) as f:
        return yaml.safe_load(f) or {}


def _resolve(node, root):
    """Tiny '${A.B.C}' resolver (absolute paths only) -- the subset of OmegaConf interpolation the
    forward-path keys use (configs/models/dd3d.yaml:1,3,53-54,59)."""
    for k, v in list(node.items()):
        if isinstance(v, dict):
            _resolve(v, root)
        elif isinstance(v, str) and v.startswith("${") and v.endswith("}"):
            cur = root
            for part in v[2:-1].split("."):
                cur = cur[part]
            node[k] = copy.deepcopy(cur)


def get_cfg(experiment="dd3d_kitti_dla34", overrides=None):
    """Return the resolved config for one of the reference's experiments
    (configs/experiments/<experiment>.yaml): base.yaml merged with the experiment delta."""
    cfg = CfgNode(_load_yaml("base.yaml"))
    cfg.merge(_load_yaml(experiment + ".yaml"))
    if overrides:
        cfg.merge(overrides)
    _resolve(cfg, cfg)
    return cfg
