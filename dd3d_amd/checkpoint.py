"""Weight loading with the behaviour of fvcore.common.checkpoint.Checkpointer [ext], which the reference uses at
scripts/train.py:50-52 (`Checkpointer(model).load(cfg.MODEL.CKPT)`) to read the released `*-remapped.pth` files
(configs/experiments/dd3d_kitti_dla34.yaml:11): the file is a pickled dict whose "model" entry is the state dict; loading is
non-strict; a "module." prefix shared by every key is stripped; numpy arrays become tensors; checkpoint entries whose shape differs
from the model's parameter are skipped with a warning instead of raising; missing / unexpected keys are logged and returned.
No network access: `path` must be a local file (the reference downloads S3 / https paths first, tridet/utils/s3.py:20-40).
"""
import logging
import os
from collections import namedtuple

import numpy as np
import torch

LOG = logging.getLogger(__name__)

IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys", "incorrect_shapes"])


def _strip_prefix_if_present(state_dict, prefix):
    keys = sorted(state_dict.keys())
    if not keys or not all(k.startswith(prefix) for k in keys):
        return
    for k in keys:
        state_dict[k[len(prefix):]] = state_dict.pop(k)


class Checkpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=True, **checkpointables):
        self.model = model
        self.save_dir = save_dir
        self.save_to_disk = save_to_disk
        self.checkpointables = dict(checkpointables)

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(kwargs)
        path = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, path)
        return path

    def load(self, path, checkpointables=None):
        """Returns what is left of the checkpoint dict after the model (and the requested checkpointables) took their entries; the
        key report of the last load is kept in `self.incompatible`."""
        if not path:
            LOG.info("No checkpoint found. Initializing model from scratch")
            return {}
        if "://" in path and not path.startswith("file://"):
            raise FileNotFoundError(f"{path}: remote checkpoints must be downloaded first (no network access in this build)")
        path = path[len("file://"):] if path.startswith("file://") else path
        if not os.path.isfile(path):
            raise AssertionError(f"Checkpoint {path} not found!")
        checkpoint = torch.load(path, map_location=torch.device("cpu"), weights_only=False)
        if "model" not in checkpoint:  # a bare state dict
            checkpoint = {"model": checkpoint}
        self.incompatible = self._load_model(checkpoint)
        for key in self.checkpointables if checkpointables is None else checkpointables:
            if key in checkpoint:
                self.checkpointables[key].load_state_dict(checkpoint.pop(key))
        return checkpoint

    def _load_model(self, checkpoint):
        sd = checkpoint.pop("model")
        for k, v in list(sd.items()):
            if isinstance(v, np.ndarray):
                sd[k] = torch.from_numpy(v)
            elif not isinstance(v, torch.Tensor):
                raise ValueError(f"Unsupported type found in checkpoint! {k}: {type(v)}")
        _strip_prefix_if_present(sd, "module.")
        model_sd = self.model.state_dict()
        incorrect = []
        for k in list(sd.keys()):
            if k in model_sd and tuple(model_sd[k].shape) != tuple(sd[k].shape):
                incorrect.append((k, tuple(sd[k].shape), tuple(model_sd[k].shape)))
                sd.pop(k)
        r = self.model.load_state_dict(sd, strict=False)
        missing = [k for k in r.missing_keys if k not in {i[0] for i in incorrect}]
        for k, s_ckpt, s_model in incorrect:
            LOG.warning("Skip loading parameter '%s' to the model due to incompatible shapes: %s in the checkpoint but %s in the model!", k,
                        s_ckpt, s_model)
        if missing:
            LOG.warning("Some model parameters or buffers are not found in the checkpoint:\n%s", "\n".join(missing))
        if r.unexpected_keys:
            LOG.warning("The checkpoint state_dict contains keys that are not used by the model:\n%s", "\n".join(r.unexpected_keys))
        return IncompatibleKeys(missing, list(r.unexpected_keys), incorrect)
