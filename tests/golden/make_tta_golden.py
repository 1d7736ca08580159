"""Golden vectors for the TTA wrapper: the reference's own DD3DWithTTA (tridet/modeling/dd3d/test_time_augmentation.py) around the
reference's own DD3D, on CPU, on top of the third-party shims of ref_shims.py (detectron2 / fvcore transforms restated there, PIL real).

    python tests/golden/make_tta_golden.py        ->  tests/golden/tta_dla34.npz            (toy scales, seconds)
    python tests/golden/make_tta_golden.py nusc   ->  tests/golden/tta_nusc_dla34.npz       (toy scales)
    python tests/golden/make_tta_golden.py nusc full -> tests/golden/tta_nusc_dla34_scales.npz
        the nuScenes experiment's own TTA (configs/experiments/dd3d_nusc_dla34.yaml:55-62: MIN_SIZES [640, 768, 896, 1024, 1152] x flip,
        IMS_PER_BATCH 96) on one 6-camera sample of raw 900 x 1600 frames: per camera ten copies forwarded as ONE batch on a 1152 x 2048
        canvas (nuscenes_dd3d_tta.py:21-178), then the sample-level aggregation (~10 min of CPU, ~25 GB)
    python tests/golden/make_tta_golden.py full   ->  tests/golden/tta_dla34_kitti_scales.npz
        the experiment's own TTA (configs/experiments/dd3d_kitti_dla34.yaml:44-53: MIN_SIZES [320, 384, 448, 512, 576] x flip, IMS_PER_BATCH 80)
        on one raw-KITTI-sized 370 x 1224 frame: ten augmented copies forwarded as ONE batch on a 640 x 1920 canvas (~2 min of CPU)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.golden import ref_shims  # noqa: E402
from tests.golden.make_golden import TRAINING_ONLY_KEYS, _merge  # noqa: E402

TTA_OVERRIDES = {
    "DD3D": {"INFERENCE": {"DO_POSTPROCESS": False, "DO_BEV_NMS": True}, "FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.02}}},
    "TEST": {"IMS_PER_BATCH": 2, "AUG": {"ENABLED": True, "MIN_SIZES": [96, 128, 160], "MAX_SIZE": 100000, "FLIP": True}},
    "INPUT": {"FORMAT": "BGR"},
}


def tta_case():
    from dd3d_amd.synthetic import KITTI_K
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, (3, 110, 260), dtype=np.uint8)
    K = torch.tensor(KITTI_K) * torch.tensor([[260 / 1224], [110 / 370], [1.0]])
    return {"image": torch.from_numpy(raw), "intrinsics": K, "height": 110, "width": 260}


# The experiment as it is run (scripts/train.py:197-228 do_test with TEST.AUG.ENABLED): only the flag do_test itself flips
# (postprocess_in_inference = False, train.py:206-209) and the input format; sizes, flip, batch size, thresholds are the experiment's.
FULL_TTA_OVERRIDES = {"DD3D": {"INFERENCE": {"DO_POSTPROCESS": False}}, "INPUT": {"FORMAT": "BGR"}}


def full_tta_case():
    """One synthetic frame of raw KITTI size (370 x 1224) with the cam-2 intrinsics."""
    from dd3d_amd.synthetic import KITTI_K
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, (3, 370, 1224), dtype=np.uint8)
    return {"image": torch.from_numpy(raw), "intrinsics": torch.tensor(KITTI_K), "height": 370, "width": 1224}


NUSC_TTA_OVERRIDES = {
    "DD3D": {"INFERENCE": {"DO_POSTPROCESS": False}, "FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.01}},
             "NUSC": {"INFERENCE": {"MAX_NUM_DETS_PER_SAMPLE": 120}}},
    "TEST": {"IMS_PER_BATCH": 2, "AUG": {"ENABLED": True, "MIN_SIZES": [96, 128], "MAX_SIZE": 100000, "FLIP": True}},
    "INPUT": {"FORMAT": "BGR"},
}


def nusc_tta_case():
    """One 6-camera sample of raw 100 x 178 frames (camera poses of dd3d_amd.synthetic)."""
    from dd3d_amd.synthetic import NUSC_K, make_inputs
    base = make_inputs(6, 100, 178, dataset="nusc", seed=2000)
    K = torch.tensor(NUSC_K) * torch.tensor([[178 / 1600], [100 / 900], [1.0]])
    return [dict(x, intrinsics=K.clone(), height=100, width=178) for x in base]


# The nuScenes experiment as it is run: only what do_test flips (postprocess_in_inference = False, scripts/train.py:206-209) and the
# input format; MIN_SIZES [640 .. 1152] x flip, IMS_PER_BATCH 96, thresholds and the per-sample cap are the experiment's.
NUSC_FULL_TTA_OVERRIDES = {"DD3D": {"INFERENCE": {"DO_POSTPROCESS": False}}, "INPUT": {"FORMAT": "BGR"}}


def nusc_full_tta_case():
    """One 6-camera sample of raw nuScenes-sized 900 x 1600 frames with the CAM_FRONT intrinsics and the synthetic camera poses."""
    from dd3d_amd.synthetic import NUSC_K, make_inputs
    base = make_inputs(6, 900, 1600, dataset="nusc", seed=2100)
    return [dict(x, intrinsics=torch.tensor(NUSC_K), height=900, width=1600) for x in base]


def nusc_main(full=False):
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.synthetic import load_calib, make_state_dict
    cfg = get_cfg("dd3d_nusc_dla34", _merge(TRAINING_ONLY_KEYS, NUSC_FULL_TTA_OVERRIDES if full else NUSC_TTA_OVERRIDES))
    sd = make_state_dict(META_ARCH_REGISTRY.get("NuscenesDD3D")(cfg), calib=load_calib("dla34_nusc"))
    ref_shims.install()
    for pkg in ("tridet.data", "tridet.data.augmentations"):
        m = sys.modules.get(pkg) or types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref_shims.REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    importlib.import_module("tridet.data.augmentations.flip_transform")
    importlib.import_module("tridet.data.augmentations.resize_transform")
    from tridet.modeling.dd3d.nuscenes_dd3d import NuscenesDD3D
    from tridet.modeling.dd3d.nuscenes_dd3d_tta import NuscenesDD3DWithTTA
    from tridet.structures.pose import Pose as RefPose
    ref = NuscenesDD3D(cfg)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    xs = nusc_full_tta_case() if full else nusc_tta_case()
    for x in xs:
        x["pose"] = RefPose(wxyz=x["pose"].quat.elements, tvec=x["pose"].tvec)
    import time
    t0 = time.perf_counter()
    out = {}
    with torch.no_grad():
        wrapper = NuscenesDD3DWithTTA(cfg, ref)
        if full:  # what went INTO the per-camera merge: the augmented copies' shapes and detection counts (the wrapper's own mapper / _batch_inference)
            aug = wrapper.tta_mapper(dict(xs[0]))
            out["copy_shapes"] = np.array([tuple(a["image"].shape[1:]) for a in aug])
            out["batch_size"] = np.array(wrapper.batch_size)
            merged_per_image = []
            inner = wrapper._inference_one_image

            def _counting(x, inner=inner):
                r = inner(x)
                merged_per_image.append(len(r))
                print(f"  camera {len(merged_per_image) - 1}: {len(r)} merged detections, {time.perf_counter() - t0:.0f} s", flush=True)
                return r

            wrapper._inference_one_image = _counting
        res = wrapper(xs)
        if full:
            out["merged_per_image"] = np.array(merged_per_image)
    print(f"reference NuscenesDD3DWithTTA: {time.perf_counter() - t0:.1f} s")
    for i, r in enumerate(res):
        inst = r["instances"]
        out[f"n{i}"] = np.array(len(inst))
        if len(inst) == 0:
            continue
        out[f"boxes{i}"], out[f"scores_3d{i}"] = inst.pred_boxes.tensor.numpy(), inst.scores_3d.numpy()
        out[f"classes{i}"], out[f"attributes{i}"], out[f"speeds{i}"] = inst.pred_classes.numpy(), inst.pred_attributes.numpy(), inst.pred_speeds.numpy()
        out[f"vectorize{i}"], out[f"global{i}"] = inst.pred_boxes3d.vectorize().numpy(), inst.pred_boxes3d_global.vectorize().numpy()
    path = os.path.join(HERE, "tta_nusc_dla34_scales.npz" if full else "tta_nusc_dla34.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, [int(out[f"n{i}"]) for i in range(len(res))])


def main(full=False):
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.synthetic import load_calib, make_state_dict
    cfg = get_cfg("dd3d_kitti_dla34", _merge(TRAINING_ONLY_KEYS, FULL_TTA_OVERRIDES if full else TTA_OVERRIDES))
    sd = make_state_dict(META_ARCH_REGISTRY.get("DD3D")(cfg), calib=load_calib("dla34_kitti"))
    ref_shims.install()
    for pkg in ("tridet.data", "tridet.data.augmentations"):
        m = sys.modules.get(pkg) or types.ModuleType(pkg)
        m.__path__ = [os.path.join(ref_shims.REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    importlib.import_module("tridet.data.augmentations.flip_transform")  # registers the intrinsics / box3d transform types
    importlib.import_module("tridet.data.augmentations.resize_transform")
    from tridet.modeling.dd3d.core import DD3D
    from tridet.modeling.dd3d.test_time_augmentation import DD3DWithTTA
    from tridet.structures.pose import Pose as RefPose
    ref = DD3D(cfg)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    x = full_tta_case() if full else tta_case()
    x["extrinsics"] = RefPose()  # DO_BEV_NMS inside every augmented forward reads a pose (core.py:138-141)
    import time
    t0 = time.perf_counter()
    with torch.no_grad():
        wrapper = DD3DWithTTA(cfg, ref)
        if full:  # also record what went INTO the merge: detections per augmented copy (the wrapper's own _batch_inference)
            aug = wrapper.tta_mapper(dict(x))
            per_copy = [len(o) for o in wrapper._batch_inference([{k: v for k, v in a.items() if k != "transforms"} for a in aug])]
            shapes = [tuple(a["image"].shape[1:]) for a in aug]
        inst = wrapper([x])[0]["instances"]
    print(f"reference DD3DWithTTA: {time.perf_counter() - t0:.1f} s")
    out = dict(boxes=inst.pred_boxes.tensor.numpy(), scores=inst.scores.numpy(), scores_3d=inst.scores_3d.numpy(), classes=inst.pred_classes.numpy(),
               vectorize=inst.pred_boxes3d.vectorize().numpy(), proj_ctr=inst.pred_boxes3d.proj_ctr.numpy(), depth=inst.pred_boxes3d.depth.numpy(),
               image_size=np.array(inst.image_size))
    if full:
        out.update(per_copy=np.array(per_copy), copy_shapes=np.array(shapes), batch_size=np.array(wrapper.batch_size))
        print("augmented copies", shapes, "detections per copy", per_copy)
    path = os.path.join(HERE, "tta_dla34_kitti_scales.npz" if full else "tta_dla34.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(inst), "merged detections")


if __name__ == "__main__":
    if "nusc" in sys.argv[1:]:
        nusc_main(full="full" in sys.argv[1:])
    else:
        main(full="full" in sys.argv[1:])
