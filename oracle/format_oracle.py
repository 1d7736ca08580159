"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's result formatting; never imported by the product.

Follows tridet/evaluators/kitti_3d_evaluator.py:86-148 (`process`), :205-264 (`convert_3d_box_to_kitti`) and
tridet/evaluators/nuscenes_evaluator.py:147-247 (`process`, `build_nusc_detection`), one box at a time, float64 numpy.

Third-party arithmetic restated here (not under /root/reference, not installed; parity unpinned for these pieces):
pyquaternion 0.9.x (`Quaternion(axis=, radians=)`, `__mul__` = `_q_matrix() @ q`, `inverse`, `_normalise` with the 1e-14 unit test,
`axis`, `angle` with `_wrap_angle`, `rotation_matrix` = `(Q Qbar^T)[1:, 1:]`) and detectron2 `BoxMode.convert` for XYXY<->XYWH lists.
Pinned by tests/golden/format_results.json, produced by the reference's own functions run over the pyquaternion restatement in
tests/golden/ref_shims.py (tests/golden/make_format_golden.py).
"""
import math
from collections import OrderedDict, defaultdict

import numpy as np


# ---- pyquaternion [ext] -------------------------------------------------------------------------------------------------------
def q_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    mag_sq = float(axis @ axis)
    if abs(1.0 - mag_sq) > 1e-12:
        axis = axis / math.sqrt(mag_sq)
    return np.concatenate([[math.cos(angle / 2.0)], axis * math.sin(angle / 2.0)])


def q_matrix(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])


def q_bar_matrix(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])


def q_mul(a, b):
    return q_matrix(a) @ b


def q_inverse(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0]) / float(q @ q)


def q_normalised(q):
    ss = float(q @ q)
    if not abs(1.0 - ss) < 1e-14 and ss > 0:
        return q / math.sqrt(ss)
    return q


def q_axis(q):
    q = q_normalised(q)
    n = np.linalg.norm(q[1:])
    return np.zeros(3) if n < 1e-17 else q[1:] / n


def q_angle(q):
    q = q_normalised(q)
    theta = 2.0 * math.atan2(np.linalg.norm(q[1:]), q[0])
    r = ((theta + math.pi) % (2 * math.pi)) - math.pi
    return math.pi if r == -math.pi else r


def q_rotation_matrix(q):
    q = q_normalised(q)
    return (q_matrix(q) @ q_bar_matrix(q).T)[1:, 1:]


# ---- kitti_3d_evaluator.py:205-264 ----------------------------------------------------------------------------------------------
def convert_3d_box_to_kitti(vec10):
    """vec10 = float32 (quat wxyz, tvec, size WLH) of one box -> (W, L, H, x, y, z, rot_y, alpha)."""
    vec10 = np.asarray(vec10, dtype=np.float32)
    quat = np.asarray(vec10[:4].tolist(), dtype=np.float64)
    tvec = vec10[4:7].copy()
    sizes = vec10[7:10]
    tvec += np.array([0., sizes[2] / 2.0, 0])  # in place on float32
    quat = q_mul(q_inverse(q_from_axis_angle([1, 0, 0], np.pi / 2)), quat)
    rot_y = -q_angle(quat) if q_axis(quat)[2] > 0 else q_angle(quat)
    # kitti_pose * [[0,0,1],[0,0,0]] then [:, ::2]: row 1 is the translation's (x, z)
    tx, tz = float(tvec[0]), float(tvec[2])
    theta = np.arctan2(abs(tx), abs(tz))
    alpha = rot_y + theta if tx < 0 else rot_y - theta
    if alpha > np.pi:
        alpha -= 2.0 * np.pi
    elif alpha < -np.pi:
        alpha += 2.0 * np.pi
    alpha = np.around(alpha, decimals=2)
    return sizes[0], sizes[1], sizes[2], tvec[0], tvec[1], tvec[2], rot_y, alpha


def xyxy_to_xywh(box):
    a = np.array(box, dtype=np.float64)
    a[2] -= a[0]
    a[3] -= a[1]
    return a.tolist()


def kitti_process(inputs, outputs, class_names, dataset_dicts):
    """kitti_3d_evaluator.py:86-148 on plain numpy outputs: each output = dict(pred_classes, pred_boxes (n,4), box3d_vec (n,10),
    scores, scores_3d).  Returns (predictions_as_json, predictions rows per image, ground-truth rows per image)."""
    as_json, pred_rows, gt_rows = [], [], []
    by_name = {d["file_name"]: d for d in dataset_dicts}
    for inp, o in zip(inputs, outputs):
        rows = []
        for c, vec, s3, b, s in zip(o["pred_classes"], o["box3d_vec"], o["scores_3d"], o["pred_boxes"], o["scores"]):
            name = class_names[int(c)]
            as_json.append(OrderedDict(category_id=int(c), category=name, bbox3d=np.asarray(vec, np.float32).tolist(),
                                       bbox=xyxy_to_xywh(np.asarray(b, np.float32).tolist()), score=float(s), score_3d=float(s3),
                                       file_name=inp["file_name"], image_id=inp["image_id"]))
            W, L, H, x, y, z, rot_y, alpha = convert_3d_box_to_kitti(vec)
            l, t, r, bb = np.asarray(b, np.float32).tolist()
            rows.append([name, -1, -1, alpha, l, t, r, bb, H, W, L, x, y, z, rot_y, float(s3)])
        pred_rows.append(rows)
        gt = by_name[inp["file_name"]]
        if "annotations" not in gt:
            continue
        rows = []
        for anno in gt["annotations"]:
            W, L, H, x, y, z, rot_y, alpha = convert_3d_box_to_kitti(np.asarray(anno["bbox3d"], np.float32))
            l, t, r, bb = anno["bbox"]  # XYXY_ABS in the fixtures
            rows.append([class_names[anno["category_id"]], -1, -1, alpha, l, t, r, bb, H, W, L, x, y, z, rot_y])
        gt_rows.append(rows)
    return as_json, pred_rows, gt_rows


# ---- nuscenes_evaluator.py:147-247 ------------------------------------------------------------------------------------------------
CATEGORIES = ["barrier", "bicycle", "bus", "car", "construction_vehicle", "motorcycle", "pedestrian", "traffic_cone", "trailer", "truck"]
VEH, PED, CYC = ("car", "bus", "construction_vehicle", "trailer", "truck"), ("pedestrian", ), ("bicycle", "motorcycle")
VEH_NAMES = {0: "vehicle.moving", 1: "vehicle.parked", 2: "vehicle.stopped"}
PED_NAMES = {0: "pedestrian.moving", 1: "pedestrian.standing", 2: "pedestrian.sitting_lying_down"}
CYC_NAMES = {0: "cycle.with_rider", 1: "cycle.without_rider"}


def nusc_process(inputs, outputs, num_images_per_sample=6):
    """Each output = dict(pred_classes, pred_boxes, box3d_vec, box3d_global_vec, scores, scores_3d, pred_attributes, pred_speeds)."""
    tokens = [x["sample_token"] for x in inputs]
    groups = defaultdict(list)
    for i, t in enumerate(tokens):
        groups[t].append(i)
    if not all(len(v) == num_images_per_sample for v in groups.values()):
        raise ValueError("Group sizes does not match with 'num_images_per_sample'.")
    as_json, results = [], defaultdict(list)
    for t in set(tokens):
        results[t]  # noqa
    for idx, (inp, o) in enumerate(zip(inputs, outputs)):
        for c, vec, g, s3, b, attr, speed, s in zip(o["pred_classes"], o["box3d_vec"], o["box3d_global_vec"], o["scores_3d"], o["pred_boxes"],
                                                    o["pred_attributes"], o["pred_speeds"], o["scores"]):
            name = CATEGORIES[int(c)]
            attr = int(attr)
            if name in VEH:
                attr_name = VEH_NAMES[attr % 3]
            elif name in PED:
                attr_name = PED_NAMES[attr % 3]
            elif name in CYC:
                attr_name = CYC_NAMES[attr % 2]
            else:
                attr_name = ""
            g = np.asarray(g, np.float32)
            vel = np.float32(speed) * q_rotation_matrix(np.asarray(g[:4].tolist(), np.float64)).T[0]
            vx, vy = vel[:2].tolist()
            as_json.append(OrderedDict(category_id=int(c), category=name, bbox3d=np.asarray(vec, np.float32).tolist(),
                                       bbox=xyxy_to_xywh(np.asarray(b, np.float32).tolist()), score=float(s), score_3d=float(s3),
                                       file_name=inp["file_name"], image_id=inp["image_id"]))
            gl = g.tolist()
            results[tokens[idx]].append({"sample_token": tokens[idx], "rotation": gl[:4], "translation": gl[4:7], "size": gl[7:],
                                         "detection_name": name, "detection_score": float(np.float32(s3)), "attribute_name": attr_name,
                                         "velocity": [vx, vy]})
    return as_json, dict(results)
