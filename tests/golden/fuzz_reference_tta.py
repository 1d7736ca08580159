"""Dev tool (needs /root/reference): the TTA oracle against the reference's own DD3DWithTTA over varied TEST.AUG settings (scale sets,
a MAX_SIZE that clamps, flip on/off, batch size of the augmented forwards, BEV NMS on/off, raw image sizes); no fixtures written.

    python tests/golden/fuzz_reference_tta.py [n_cases] [seed]
"""
import importlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.golden import ref_shims  # noqa: E402
from tests.golden.make_golden import TRAINING_ONLY_KEYS, _merge  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.structures import Pose
    from dd3d_amd.synthetic import KITTI_K, load_calib, make_state_dict
    from oracle import tta_oracle as T
    ref_shims.install()
    for pkg in ("tridet.data", "tridet.data.augmentations"):
        m = sys.modules.get(pkg) or types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref_shims.REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    importlib.import_module("tridet.data.augmentations.flip_transform")
    importlib.import_module("tridet.data.augmentations.resize_transform")
    from tridet.modeling.dd3d.core import DD3D
    from tridet.modeling.dd3d.test_time_augmentation import DD3DWithTTA
    from tridet.structures.pose import Pose as RefPose
    for it in range(n):
        pick = lambda *xs: rng.choice(xs)  # noqa: E731
        aug = {"ENABLED": True, "MIN_SIZES": pick([96, 128, 160], [128], [64, 192], [100, 140]), "MAX_SIZE": pick(100000, 300, 220),
               "FLIP": pick(True, False)}
        over = {"DD3D": {"INFERENCE": {"DO_POSTPROCESS": False, "DO_BEV_NMS": pick(True, False)},
                         "FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": pick(0.02, 0.05)}}},
                "TEST": {"IMS_PER_BATCH": pick(1, 3, 4), "AUG": aug}, "INPUT": {"FORMAT": "BGR"}}
        h, w = pick((110, 260), (96, 200), (130, 310))
        cfg = get_cfg("dd3d_kitti_dla34", _merge(dict(TRAINING_ONLY_KEYS), over))
        sd = make_state_dict(META_ARCH_REGISTRY.get("DD3D")(cfg), calib=load_calib("dla34_kitti"))
        ref = DD3D(cfg)
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        raw = torch.from_numpy(np.random.default_rng(it).integers(0, 256, (3, h, w), dtype=np.uint8))
        K = torch.tensor(KITTI_K) * torch.tensor([[w / 1224], [h / 370], [1.0]])
        x_ref = {"image": raw.clone(), "intrinsics": K.clone(), "height": h, "width": w, "extrinsics": RefPose()}
        x = {"image": raw.clone(), "intrinsics": K.clone(), "height": h, "width": w, "extrinsics": Pose()}
        with torch.no_grad():
            inst = DD3DWithTTA(cfg, ref)([x_ref])[0]["instances"]
            r = T.tta_forward(sd, cfg, x)
        assert len(inst) == len(r["scores_3d"]), (len(inst), len(r["scores_3d"]), over)
        if len(inst):
            assert torch.equal(inst.pred_classes, r["pred_classes"])
            assert torch.allclose(inst.pred_boxes.tensor, r["pred_boxes"], rtol=1e-5, atol=3e-3) and torch.allclose(inst.scores_3d, r["scores_3d"], rtol=1e-5, atol=1e-6)
            v, g = r["vec"], inst.pred_boxes3d.vectorize()
            assert torch.allclose(v[:, 4:], g[:, 4:], rtol=1e-5, atol=1e-3)
            assert float(torch.minimum((v[:, :4] - g[:, :4]).abs().amax(1), (v[:, :4] + g[:, :4]).abs().amax(1)).max()) < 1e-4
        print(f"[{it:2d}] ok  raw {h}x{w}  merged {len(inst):4d} of {r['n_union']:4d}  {aug}  batch {over['TEST']['IMS_PER_BATCH']}  "
              f"bev {over['DD3D']['INFERENCE']['DO_BEV_NMS']}", flush=True)
    print("all", n, "TTA cases agree with the reference")


if __name__ == "__main__":
    main()
