"""Golden vectors for the evaluator-side result formatting, produced by the REFERENCE'S OWN code:
`convert_3d_box_to_kitti`, `KITTI3DEvaluator.process` (tridet/evaluators/kitti_3d_evaluator.py) and `NuscenesEvaluator.process`
(tridet/evaluators/nuscenes_evaluator.py), imported from /root/reference and run on CPU.  Third-party modules that are not installed
are restated in ref_shims.py (pyquaternion, detectron2 structures) or replaced here by inert placeholders (numba decorators,
detectron2 catalogs / DatasetEvaluator / BoxMode, iopath, the nuScenes devkit, seaborn-dependent dataset tables).

    python tests/golden/make_format_golden.py   ->  tests/golden/format_results.json
"""
import enum
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.golden import ref_shims  # noqa: E402
from tests.golden.ref_shims import _mod  # noqa: E402

KITTI_CLASSES = ["Car", "Pedestrian", "Cyclist", "Van", "Truck"]  # configs/train_datasets/kitti_3d.yaml class order


class BoxMode(enum.IntEnum):
    """[ext] detectron2.structures.BoxMode, the two modes the formatting code uses; `convert` on a list works on a float64 numpy
    copy and returns a list."""
    XYXY_ABS = 0
    XYWH_ABS = 1

    @staticmethod
    def convert(box, from_mode, to_mode):
        single = isinstance(box, (list, tuple))
        arr = np.array(box, dtype=np.float64)[None, :] if single else np.array(box, dtype=np.float64)
        if from_mode == to_mode:
            pass
        elif from_mode == BoxMode.XYXY_ABS and to_mode == BoxMode.XYWH_ABS:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        elif from_mode == BoxMode.XYWH_ABS and to_mode == BoxMode.XYXY_ABS:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        else:
            raise NotImplementedError
        return type(box)(arr.flatten().tolist()) if single else arr


def install():
    ref_shims.install()
    ident = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))  # noqa: E731
    errors = _mod("numba.errors", NumbaDeprecationWarning=DeprecationWarning) if "numba" in sys.modules else None
    nb = _mod("numba", jit=ident, float32=np.float32)
    cuda = _mod("numba.cuda", jit=ident, select_device=lambda *_: None)
    errors = _mod("numba.errors", NumbaDeprecationWarning=DeprecationWarning)
    nb.cuda, nb.errors = cuda, errors
    sys.modules["mpi4py"].MPI = types.SimpleNamespace(COMM_WORLD=types.SimpleNamespace(Get_rank=lambda: 0))
    torch.cuda.device_count = lambda: 1  # module-level `rank % device_count()` of tridet/evaluators/rotate_iou.py
    _mod("detectron2.data.catalog", DatasetCatalog=None, MetadataCatalog=None)
    _mod("detectron2.evaluation")
    _mod("detectron2.evaluation.evaluator", DatasetEvaluator=object)
    _mod("detectron2.structures.boxes", BoxMode=BoxMode)
    _mod("iopath")
    _mod("iopath.common")
    _mod("iopath.common.file_io", PathManager=None)
    _mod("nuscenes", NuScenes=None)
    for name in ("nuscenes.eval", "nuscenes.eval.common", "nuscenes.eval.detection"):
        _mod(name)
    _mod("nuscenes.eval.common.config", config_factory=None)
    _mod("nuscenes.eval.common.data_classes", EvalBoxes=None)
    _mod("nuscenes.eval.common.loaders", add_center_dist=None, filter_eval_boxes=None, load_gt=None, load_prediction=None)
    _mod("nuscenes.eval.detection.data_classes", DetectionBox=None)
    _mod("nuscenes.eval.detection.evaluate", DetectionEval=object)
    from collections import OrderedDict
    # tridet/data/datasets/nuscenes/build.py:39-61 (the module itself needs seaborn + the devkit): tables copied as data
    build = sys.modules["tridet.data.datasets.nuscenes.build"]
    build.ATTRIBUTE_IDS = {
        'vehicle.moving': 0, 'vehicle.parked': 1, 'vehicle.stopped': 2, 'pedestrian.moving': 0, 'pedestrian.standing': 1,
        'pedestrian.sitting_lying_down': 2, 'cycle.with_rider': 0, 'cycle.without_rider': 1
    }
    build.CATEGORY_IDS = OrderedDict((n, i) for i, n in enumerate(
        ['barrier', 'bicycle', 'bus', 'car', 'construction_vehicle', 'motorcycle', 'pedestrian', 'traffic_cone', 'trailer', 'truck']))
    build.DATASET_NAME_TO_VERSION = {}
    m = _mod("tridet.evaluators")
    m.__path__ = [os.path.join(ref_shims.REFERENCE_ROOT, "tridet", "evaluators")]
    _mod("tqdm", tqdm=lambda x, *a, **k: x) if "tqdm" not in sys.modules else None


def conversion_boxes(rng, n=192):
    """(n, 10) float32 box vectors with the cases the conversion branches on."""
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = np.concatenate([rng.uniform(-30, 30, (n, 1)), rng.uniform(-2, 3, (n, 1)), rng.uniform(1, 70, (n, 1))], 1)
    s = rng.uniform(0.4, 6.0, (n, 3))
    v = np.concatenate([q, t, s], 1).astype(np.float32)
    h = np.float32(np.sqrt(0.5))
    special = [
        [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1],  # exactly unit: no renormalisation
        [h, h, 0, 0],      # cancels the inversion: zero rotation angle, undefined axis
        [h, -h, 0, 0], [h, 0, h, 0], [h, 0, -h, 0], [0.5, 0.5, 0.5, 0.5], [0.5, 0.5, -0.5, 0.5], [2, 0, 0, 0], [0.3, 0.3, 0.1, -0.2],  # not unit
    ]
    for i, sq in enumerate(special):
        v[i, :4] = sq
    # yaw-only boxes (what KITTI annotations are): q = Rx(90deg)-style upright boxes turned about the camera's y axis
    for i, yaw in enumerate(np.linspace(-np.pi, np.pi, 17)):
        qy = np.array([np.cos(yaw / 2), 0, np.sin(yaw / 2), 0])
        qx = np.array([np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0])
        w1, x1, y1, z1 = qy
        w2, x2, y2, z2 = qx
        v[20 + i, :4] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                         w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
    v[40, 4] = 0.0           # x = 0: theta = 0, the `tx < 0` branch is not taken
    v[41, 6] = -5.0          # behind the camera
    v[42, 4:7] = [-25, 1, 2]  # large viewing angle, alpha wraps
    v[43, 4:7] = [25, 1, 2]
    return v


def make_outputs(rng, n_images, dets_per_image, num_classes, nusc=False):
    from tridet.structures.boxes3d import Boxes3D, GenericBoxes3D
    from dd3d_amd.structures import Boxes, Instances
    inputs, outputs, plain = [], [], []
    for i in range(n_images):
        n = dets_per_image[i]
        q = rng.normal(size=(n, 4)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        K = np.array([[700.0, 0, 320], [0, 700.0, 180], [0, 0, 1]], dtype=np.float32)
        inv_K = torch.from_numpy(np.linalg.inv(K).astype(np.float32))
        proj = torch.from_numpy(rng.uniform(20, 600, (n, 2)).astype(np.float32))
        depth = torch.from_numpy(rng.uniform(3, 60, (n, 1)).astype(np.float32))
        size = torch.from_numpy(rng.uniform(0.5, 5, (n, 3)).astype(np.float32))
        b3 = Boxes3D(torch.from_numpy(q), proj, depth, size, inv_K[None].expand(n, 3, 3).contiguous())
        xy = rng.uniform(0, 300, (n, 2))
        boxes = torch.from_numpy(np.concatenate([xy, xy + rng.uniform(4, 200, (n, 2))], 1).astype(np.float32))
        inst = Instances((360, 640))
        inst.pred_boxes = Boxes(boxes)
        inst.pred_classes = torch.from_numpy(rng.integers(0, num_classes, n))
        inst.scores = torch.from_numpy(rng.uniform(0.05, 1, n).astype(np.float32))
        inst.scores_3d = torch.from_numpy(rng.uniform(0.01, 1, n).astype(np.float32))
        inst.pred_boxes3d = b3
        rec = {"pred_classes": inst.pred_classes.numpy(), "pred_boxes": boxes.numpy(), "box3d_vec": b3.vectorize().numpy(),
               "scores": inst.scores.numpy(), "scores_3d": inst.scores_3d.numpy()}
        inp = {"file_name": f"img_{i:04d}.png", "image_id": f"id{i}"}
        if nusc:
            qg = rng.normal(size=(n, 4)).astype(np.float32)
            qg /= np.linalg.norm(qg, axis=1, keepdims=True)
            if n:
                qg[0] = [1, 0, 0, 0]
            g = GenericBoxes3D(torch.from_numpy(qg), torch.from_numpy(rng.uniform(-500, 500, (n, 3)).astype(np.float32)), size)
            inst.pred_boxes3d_global = g
            inst.pred_attributes = torch.from_numpy(rng.integers(0, 3, n))
            inst.pred_speeds = torch.from_numpy(rng.uniform(0, 12, n).astype(np.float32))
            rec.update(box3d_global_vec=g.vectorize().numpy(), pred_attributes=inst.pred_attributes.numpy(), pred_speeds=inst.pred_speeds.numpy())
            inp["sample_token"] = f"tok{i // 6}"
        inputs.append(inp)
        outputs.append({"instances": inst})
        plain.append({k: np.asarray(v).tolist() for k, v in rec.items()})
    return inputs, outputs, plain


def main():
    install()
    import pandas as pd
    from tridet.evaluators import kitti_3d_evaluator as KE
    from tridet.evaluators import nuscenes_evaluator as NE
    from tridet.structures.boxes3d import GenericBoxes3D
    rng = np.random.default_rng(11)
    out = {}
    # 1. convert_3d_box_to_kitti over the branch table
    v = conversion_boxes(rng)
    conv = []
    for row in v:
        box = GenericBoxes3D.from_vectors([row.copy()])
        r = KE.convert_3d_box_to_kitti(box)
        conv.append([float(x) for x in r])
        assert [type(x).__name__ for x in r] == ["float32"] * 6 + ["float", "float64"], [type(x).__name__ for x in r]
    out["convert"] = {"box3d_vec": v.tolist(), "kitti": conv}
    # 2. KITTI3DEvaluator.process (ctor bypassed: it only reads the detectron2 catalogs)
    inputs, outputs, plain = make_outputs(rng, 3, [7, 0, 4], 5)
    dataset_dicts = []
    for i, inp in enumerate(inputs):
        d = {"file_name": inp["file_name"]}
        if i != 2:  # image 2 is a test-set image: no annotations
            d["annotations"] = [{"category_id": int(rng.integers(0, 5)), "bbox": rng.uniform(0, 300, 4).round(2).tolist(), "bbox_mode": BoxMode.XYXY_ABS,
                                 "bbox3d": v[20 + 3 * i + j].tolist()} for j in range(2)]
        dataset_dicts.append(d)
    ev = KE.KITTI3DEvaluator.__new__(KE.KITTI3DEvaluator)
    ev._dataset_dicts = {d["file_name"]: d for d in dataset_dicts}
    ev._class_names = KITTI_CLASSES
    ev.reset()
    ev.process(inputs, outputs)
    to_csv = lambda df: df.to_csv(sep=" ", header=False, index=False)  # noqa: E731  (prepare_kitti3d_submission's format)
    for d in dataset_dicts:
        for a in d.get("annotations", []):
            a["bbox_mode"] = int(a["bbox_mode"])
    out["kitti"] = {"inputs": inputs, "outputs": plain, "dataset_dicts": dataset_dicts, "class_names": KITTI_CLASSES,
                    "predictions_as_json": ev._predictions_as_json, "predictions_csv": [to_csv(df) for df in ev._predictions_kitti_format],
                    "groundtruth_csv": [to_csv(df) for df in ev._groundtruth_kitti_format],
                    "predictions_rows": [df.values.tolist() for df in ev._predictions_kitti_format]}
    # 3. NuscenesEvaluator.process: two samples of six cameras, one camera without detections
    inputs, outputs, plain = make_outputs(rng, 12, [3, 2, 0, 4, 1, 2, 5, 1, 1, 0, 2, 3], 10, nusc=True)
    nev = NE.NuscenesEvaluator("/nonexistent", "nusc_val", None)
    nev.reset()
    nev.process(inputs, outputs)
    out["nusc"] = {"inputs": inputs, "outputs": plain, "predictions_as_json": nev._predictions_as_json,
                   "sample_results": {k: v for k, v in nev._nusc_sample_results.items()}}
    with open(os.path.join(HERE, "format_results.json"), "w") as f:
        json.dump(out, f)
    print("wrote format_results.json:", len(conv), "conversions,", len(ev._predictions_as_json), "KITTI predictions,",
          sum(len(v) for v in nev._nusc_sample_results.values()), "nuScenes detections")


if __name__ == "__main__":
    main()
