"""CPU ORACLE for the nuScenes-specific part of the forward path (test infrastructure only; rules in dd3d_oracle.py).

Restates: NuscenesDD3D.forward inference branch (tridet/modeling/dd3d/nuscenes_dd3d.py:337-469), NuscenesInference
(:268-296), nuscenes_sample_aggregate / sample_bev_nms / get_group_idxs (tridet/modeling/dd3d/postprocessing.py:22-129),
boxes3d_to_rotated_boxes / bev_nms (tridet/layers/bev_nms.py:51-133), and the third-party arithmetic they call:
detectron2 batched_nms_rotated / nms_rotated / box_iou_rotated [ext, parity unpinned] and the pytorch3d Transform3d
compositions [ext], folded to plain matrix algebra.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle import dd3d_oracle as O

MAX_NUM_ATTRIBUTES = 3  # tridet/data/datasets/nuscenes/build.py:77
_EPS = 1e-5


# ----------------------------------------------------------------------------------------- [ext] box_iou_rotated
def _rotated_vertices(b):
    """[ext] detectron2 box_iou_rotated_utils.h get_rotated_vertices; b = (cx, cy, w, h, angle_deg), fp32."""
    cx, cy, w, h, a = [float(v) for v in b]
    theta = a * 0.01745329251
    c = float(torch.tensor(math.cos(theta), dtype=torch.float32)) * 0.5
    s = float(torch.tensor(math.sin(theta), dtype=torch.float32)) * 0.5
    p0 = (cx - s * h - c * w, cy + c * h - s * w)
    p1 = (cx + s * h - c * w, cy - c * h - s * w)
    return [p0, p1, (2 * cx - p0[0], 2 * cy - p0[1]), (2 * cx - p1[0], 2 * cy - p1[1])]


def _cross(a, b):
    return a[0] * b[1] - b[0] * a[1]


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1]


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def _intersection_points(p1, p2):
    pts = []
    v1 = [_sub(p1[(i + 1) % 4], p1[i]) for i in range(4)]
    v2 = [_sub(p2[(i + 1) % 4], p2[i]) for i in range(4)]
    for i in range(4):
        for j in range(4):
            det = _cross(v2[j], v1[i])
            if abs(det) <= 1e-14:
                continue
            v12 = _sub(p2[j], p1[i])
            t1 = _cross(v2[j], v12) / det
            t2 = _cross(v1[i], v12) / det
            if -_EPS < t1 < 1.0 + _EPS and -_EPS < t2 < 1.0 + _EPS:
                pts.append((p1[i][0] + v1[i][0] * t1, p1[i][1] + v1[i][1] * t1))
    for (pa, pb, vb) in ((p1, p2, v2), (p2, p1, v1)):  # vertices of pa inside pb
        AB, DA = vb[0], vb[3]
        ABdotAB, ADdotAD = _dot(AB, AB), _dot(DA, DA)
        for i in range(4):
            AP = _sub(pa[i], pb[0])
            APdotAB, APdotAD = _dot(AP, AB), -_dot(AP, DA)
            if APdotAB > -_EPS and APdotAD > -_EPS and APdotAB < ABdotAB + _EPS and APdotAD < ADdotAD + _EPS:
                pts.append(pa[i])
    return pts


def _convex_hull_area(pts):
    """Graham scan + shoelace fan, as convex_hull_graham / polygon_area of box_iou_rotated_utils.h."""
    n = len(pts)
    if n <= 2:
        return 0.0
    t = min(range(n), key=lambda i: (pts[i][1], pts[i][0]))
    start = pts[t]
    q = [_sub(p, start) for p in pts]
    q[0], q[t] = q[t], q[0]
    rest = q[1:]

    import functools

    def cmp(a, b):
        temp = _cross(a, b)
        if abs(temp) < 1e-6:
            return -1 if _dot(a, a) < _dot(b, b) else (1 if _dot(a, a) > _dot(b, b) else 0)
        return -1 if temp > 0 else 1

    rest.sort(key=functools.cmp_to_key(cmp))
    q = [q[0]] + rest
    k = 1
    while k < n and _dot(q[k], q[k]) <= 1e-8:
        k += 1
    if k == n:
        return 0.0
    q[1] = q[k]
    m = 2
    for i in range(k + 1, n):
        while m > 1 and _cross(_sub(q[i], q[m - 2]), _sub(q[m - 1], q[m - 2])) >= 0:
            m -= 1
        q[m] = q[i]
        m += 1
    if m <= 2:
        return 0.0
    area = 0.0
    for i in range(1, m - 1):
        area += abs(_cross(_sub(q[i], q[0]), _sub(q[i + 1], q[0])))
    return area / 2.0


def box_iou_rotated_single(b1, b2):
    """[ext] single_box_iou_rotated: both boxes shifted by the mean of their centres first."""
    a1, a2 = float(b1[2]) * float(b1[3]), float(b2[2]) * float(b2[3])
    if a1 < 1e-14 or a2 < 1e-14:
        return 0.0
    sx, sy = (float(b1[0]) + float(b2[0])) / 2.0, (float(b1[1]) + float(b2[1])) / 2.0
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))
    c1 = (f32(float(b1[0]) - sx), f32(float(b1[1]) - sy), b1[2], b1[3], b1[4])
    c2 = (f32(float(b2[0]) - sx), f32(float(b2[1]) - sy), b2[2], b2[3], b2[4])
    inter = _convex_hull_area(_intersection_points(_rotated_vertices(c1), _rotated_vertices(c2)))
    return inter / (a1 + a2 - inter)


def nms_rotated(boxes, scores, thr):
    """[ext] detectron2 nms_rotated (CPU reference): score order desc (stable here), suppress IoU > thr."""
    n = boxes.shape[0]
    order = torch.sort(scores, descending=True, stable=True)[1].tolist()
    suppressed = [False] * n
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = order[_j]
            if not suppressed[j] and box_iou_rotated_single(boxes[i], boxes[j]) > thr:
                suppressed[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms_rotated(boxes, scores, idxs, thr):
    """[ext] detectron2.layers.nms.batched_nms_rotated: centre offsets idxs * (max_c - min_c + 1), one nms_rotated."""
    if boxes.numel() == 0:
        return torch.empty((0, ), dtype=torch.int64)
    boxes = boxes.float()
    max_c = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    min_c = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.max(boxes[:, 2], boxes[:, 3]) / 2).min()
    offsets = idxs.to(boxes) * (max_c - min_c + 1)
    b = boxes.clone()
    b[:, :2] += offsets[:, None]
    return nms_rotated(b, scores, thr)


# ----------------------------------------------------------------------------------------- bev_nms.py / postprocessing.py
def boxes3d_to_rotated_boxes_global(quat, tvec, size):
    """bev_nms.py:51-96 with pose_cam_global = identity (postprocessing.py:54) and VEHICLE_TO_BEV_ROTATION
    (bev_nms.py:42-47): top-face corners [0,1,5,4] -> BEV (x,y) = (-Y, -X) -> (cx, cy, width, length, angle_deg)."""
    corners = O.boxes3d_corners(quat, tvec, size)  # boxes3d.py:47-64
    surface = corners[:, [0, 1, 5, 4], :]
    bev = torch.stack([-surface[..., 1], -surface[..., 0]], dim=-1)  # [ext] Transform3d(matrix=M.T).transform_points == M p
    length = torch.norm(bev[:, 0] - bev[:, 3], dim=1).abs()
    width = torch.norm(bev[:, 0] - bev[:, 1], dim=1).abs()
    center = torch.mean(bev[:, [0, 2]], dim=1)
    forward = bev[:, 0] - bev[:, 3]
    angle = 180. / math.pi * torch.atan2(forward[:, 0], forward[:, 1])
    return torch.stack([center[:, 0], center[:, 1], width, length, angle], dim=1)


def boxes_to_global(vec, pose_q, pose_t):
    """postprocessing.py:24-47: T_WO = T_WS o T_SO  =>  R_WO = R_WS R_SO,  t_WO = R_WS t_SO + t_WS, quat via
    matrix_to_quaternion ([ext] pytorch3d Transform3d row-vector algebra folded)."""
    quat, tvec, wlh = vec[:, :4], vec[:, 4:7], vec[:, 7:10]
    R_SO = O.quaternion_to_matrix(quat)
    R_WS = O.quaternion_to_matrix(torch.as_tensor(pose_q, dtype=torch.float32))
    t_WS = torch.as_tensor(pose_t, dtype=torch.float32)
    R_WO = torch.matmul(R_WS.unsqueeze(0), R_SO)
    t_WO = torch.matmul(tvec, R_WS.T) + t_WS
    return torch.cat([O.matrix_to_quaternion(R_WO), t_WO, wlh], dim=1)


def _pose_tuple(pose):
    """(quat wxyz, tvec) of a tridet Pose-like object (tridet/structures/pose.py: .quat.elements, .tvec) or of a tuple."""
    if isinstance(pose, (tuple, list)):
        return pose
    return ([float(v) for v in pose.quat.elements], [float(v) for v in pose.tvec])


def get_group_idxs(sample_tokens, num_images_per_sample):
    """postprocessing.py:111-129."""
    grouped = OrderedDict()
    for idx, tok in enumerate(sample_tokens):
        grouped.setdefault(tok, []).append(idx)
    if not all(len(v) == num_images_per_sample for v in grouped.values()):
        raise ValueError("Group sizes does not match with 'num_images_per_sample'.")
    return grouped


def nuscenes_sample_aggregate(instances, group_idxs, num_classes, poses, iou_threshold, include_boxes3d_global=True,
                              max_num_dets_per_sample=None):
    """postprocessing.py:58-108.  instances: list of dict (Instances fields); poses: list of (quat wxyz, tvec)."""
    num_images = len(instances)
    image_id, cat_id = [None] * num_images, [None] * num_images
    for group_idx, (_, idxs) in enumerate(group_idxs.items()):
        for idx in idxs:
            image_id[idx] = torch.ones_like(instances[idx]["pred_classes"]) * idx
            cat_id[idx] = instances[idx]["pred_classes"] + group_idx * num_classes
    glob = [boxes_to_global(O.boxes3d_vectorize(inst["pred_boxes3d"]), *pose) for inst, pose in zip(instances, poses)]
    glob = torch.cat(glob, 0)
    ids = torch.cat(cat_id)
    scores = torch.cat([x["scores_3d"] for x in instances])
    rot = boxes3d_to_rotated_boxes_global(glob[:, :4], glob[:, 4:7], glob[:, 7:])
    keep = batched_nms_rotated(rot, scores, ids, iou_threshold)
    if max_num_dets_per_sample:
        keep = keep[:max_num_dets_per_sample]  # quirk: truncates the batch-global keep list (postprocessing.py:93-94)
    allinst = O._cat_instances(instances)
    if include_boxes3d_global:
        allinst["pred_boxes3d_global"] = glob
    mask = torch.zeros(len(scores), dtype=torch.bool)
    mask[keep] = True
    img = torch.cat(image_id)
    out = []
    for i in range(num_images):
        out.append(O._index_instances(allinst, mask & (img == i)))
    return out, dict(rotated=rot, keep=keep, glob=glob)


# ----------------------------------------------------------------------------------------- NuscenesDD3D.forward
def nuscenes_dd3d_forward(sd, cfg, batched_inputs):
    """nuscenes_dd3d.py:337-469, inference branch.  Inputs carry 'pose' = (quat wxyz, tvec) and 'sample_token'."""
    stages = {}
    x, image_sizes, intrinsics = O.preprocess(sd, batched_inputs, O.size_divisibility(cfg))
    stages["images"] = x
    features, strides, _ = O.dd3d_backbone(sd, cfg, x)
    locations = [
        O.compute_features_locations(f.shape[-2], f.shape[-1], s, cfg["DD3D"]["FEATURE_LOCATIONS_OFFSET"])
        for f, s in zip(features, strides)
    ]
    logits, box2d_reg, centerness, cls_tower_out = O.fcos2d_head(sd, features, cfg["DD3D"]["FCOS2D"]["NUM_CLS_CONVS"],
                                                                     num_box_convs=cfg["DD3D"]["FCOS2D"]["NUM_BOX_CONVS"])
    quat, ctr, depth, size, conf = O.fcos3d_head(sd, features, cfg["DD3D"]["FCOS3D"]["NUM_CONVS"])
    attr = [O.conv2d(sd, "attr_logits", t, padding=1) for t in cls_tower_out]  # nuscenes_dd3d.py:371-374
    speed = [F.relu(O.conv2d(sd, "speed", t, padding=1)) for t in cls_tower_out]
    heads = dict(logits=logits, box2d_reg=box2d_reg, centerness=centerness, quat=quat, ctr=ctr, depth=depth, size=size, conf=conf,
                 attr=attr, speed=speed, features=features)
    stages.update(heads)
    inv_K = intrinsics.inverse()
    results, st2 = nuscenes_postprocess_from_heads(cfg, heads, locations, inv_K, image_sizes, batched_inputs)
    stages.update(st2)
    return results, stages


def nuscenes_postprocess_from_heads(cfg, heads, locations, inv_K, image_sizes, batched_inputs, sample_aggregate=True):
    """The post-head part of NuscenesDD3D.forward; with sample_aggregate=False and no attr/speed maps it is DD3D.forward's
    (core.py:114-164) including the DO_BEV_NMS branch."""
    L, B = len(heads["logits"]), len(image_sizes)
    pred, infos = [], []
    for l in range(L):
        r, info = O.fcos2d_inference_level(heads["logits"][l], heads["box2d_reg"][l], heads["centerness"][l], locations[l], cfg)
        for inst in r:
            inst["fpn_levels"] = torch.ones(len(inst["scores"]), dtype=torch.long) * l
        O.fcos3d_inference_level(heads["quat"][l], heads["ctr"][l], heads["depth"][l], heads["size"][l], heads["conf"][l], inv_K, r, info, cfg)
        infos.append(info)
        if "attr" not in heads:
            pred.append(r)
            continue
        # NuscenesInference (nuscenes_dd3d.py:268-296)
        a = heads["attr"][l].permute(0, 2, 3, 1).reshape(B, -1, MAX_NUM_ATTRIBUTES)
        s = heads["speed"][l].permute(0, 2, 3, 1).reshape(B, -1)
        for i in range(B):
            fg, tk = info[i]["fg_inds"], info[i]["topk_indices"]
            a_i, s_i = a[i][fg], s[i][fg]
            if tk is not None:
                a_i, s_i = a_i[tk], s_i[tk]
            r[i]["pred_attributes"] = a_i.argmax(dim=1) if len(a_i) else torch.zeros(0, dtype=torch.long)
            r[i]["pred_speeds"] = s_i
        pred.append(r)
    per_image = [O._cat_instances([pred[l][i] for l in range(L)]) for i in range(B)]
    stages = {"candidates": per_image, "level_info": infos}
    inf = cfg["DD3D"]["INFERENCE"]
    if inf["DO_NMS"]:
        per_image = [O.nms_and_top_k(x, cfg, "scores_3d") for x in per_image]
    if inf["DO_BEV_NMS"] or (sample_aggregate and inf["DO_POSTPROCESS"]):
        poses = [_pose_tuple(x["pose"] if "pose" in x else x["extrinsics"]) for x in batched_inputs]  # core.py:138-141
    if inf["DO_BEV_NMS"]:
        per_image, _ = nuscenes_sample_aggregate(
            per_image, OrderedDict((i, [i]) for i in range(B)), cfg["DD3D"]["NUM_CLASSES"], poses, inf["BEV_NMS_IOU_THRESH"],
            include_boxes3d_global=False
        )
    if not inf["DO_POSTPROCESS"]:
        return per_image, stages
    per_image = [
        O.detector_postprocess(inst, isz, inp.get("height", isz[0]), inp.get("width", isz[1]))
        for inst, inp, isz in zip(per_image, batched_inputs, image_sizes)
    ]
    stages["before_aggregate"] = per_image
    if not sample_aggregate:
        return per_image, stages
    nus = cfg["DD3D"]["NUSC"]["INFERENCE"]
    groups = get_group_idxs([x["sample_token"] for x in batched_inputs], nus["NUM_IMAGES_PER_SAMPLE"])
    out, agg = nuscenes_sample_aggregate(
        per_image, groups, cfg["DD3D"]["NUM_CLASSES"], poses, inf["BEV_NMS_IOU_THRESH"],
        max_num_dets_per_sample=nus["MAX_NUM_DETS_PER_SAMPLE"]
    )
    stages["aggregate"] = agg
    return out, stages
