"""CPU ORACLE for the test-time-augmentation wrapper (test infrastructure only).

Restates tridet/modeling/dd3d/test_time_augmentation.py: DatasetMapperTTA.__call__ (:37-86), DD3DWithTTA._batch_inference (:120-135),
_inference_one_image (:157-190), _get_augmented_instances (:197-260); the transforms it relies on: detectron2 ResizeShortestEdge /
ResizeTransform / HFlipTransform / TransformList.inverse / Transform.apply_box [ext], and the in-repo extensions
apply_imresize_intrinsics (resize_transform.py:13-21), apply_hflip_intrinsics / apply_hflip_box3d (flip_transform.py:7-53),
Boxes3D.from_vectors (boxes3d.py:176-217), bev_nms with CAMERA_TO_VEHICLE_ROTATION (bev_nms.py:27-47,99-133).
Pinned by tests/golden/tta_dla34.npz, produced by the reference wrapper itself (tests/golden/make_tta_golden.py).
"""
import numpy as np
import torch

from oracle import dd3d_oracle as O
from oracle import nuscenes_oracle as N
from oracle import resize_oracle as R

CAM_TO_VEHICLE_QUAT = (0.5, -0.5, 0.5, -0.5)  # rotation [[0,0,1],[-1,0,0],[0,-1,0]]


def augment(x, min_sizes, max_size, flip):
    """:37-86 -> list of (input dict, (resize (h, w, nh, nw), flip width or None))."""
    img = x["image"].numpy()
    _, h, w = img.shape
    assert (h, w) == (x["height"], x["width"]), "pre-transform (input already resized) is not exercised by the oracle"
    out = []
    for s in min_sizes:
        nh, nw = R.shortest_edge_size(h, w, s, max_size)
        resized = R.resize_bilinear_u8(img, nh, nw)
        for f in ((False, True) if flip else (False, )):
            d = {k: v for k, v in x.items() if k != "image"}
            K = R.resize_intrinsics(x["intrinsics"].numpy(), h, w, nh, nw)
            im = resized
            if f:
                im = np.ascontiguousarray(resized[:, :, ::-1])
                K[0, 2] = nw - K[0, 2]
            d["image"], d["intrinsics"] = torch.from_numpy(np.ascontiguousarray(im)), torch.from_numpy(K)
            out.append((d, ((h, w, nh, nw), nw if f else None)))
    return out


def tta_forward(sd, cfg, x):
    """One image through DD3DWithTTA.  cfg must have DO_POSTPROCESS False.  Returns the merged instances dict (+ intermediates)."""
    inf, aug = cfg["DD3D"]["INFERENCE"], cfg["TEST"]["AUG"]
    assert not inf["DO_POSTPROCESS"]
    copies = augment(x, aug["MIN_SIZES"], aug["MAX_SIZE"], aug["FLIP"])
    bs = cfg["TEST"]["IMS_PER_BATCH"]
    outputs = []
    for i in range(0, len(copies), bs):  # :120-135
        res, _ = O.dd3d_forward(sd, cfg, [c[0] for c in copies[i:i + bs]])
        outputs += res
    boxes, vecs, pcs, scores, scores3d, classes = [], [], [], [], [], []
    for (d, ((h, w, nh, nw), fw)), o in zip(copies, outputs):
        b = o["pred_boxes"].numpy().astype(np.float32)
        xs, ys = b[:, [0, 2, 0, 2]].copy(), b[:, [1, 1, 3, 3]].copy()
        v = O.boxes3d_vectorize(o["pred_boxes3d"]).numpy().astype(np.float32)
        K = d["intrinsics"].numpy().astype(np.float32).copy()
        if fw is not None:  # inverse flip first (TransformList.inverse)
            xs = fw - xs
            v = np.concatenate([v[:, [3]], -v[:, [2]], -v[:, [1]], v[:, [0]], -v[:, 4:5], v[:, 5:7], v[:, 7:]], axis=1)
            K[0, 2] = fw - K[0, 2]
        xs, ys = xs * (w * 1.0 / nw), ys * (h * 1.0 / nh)
        K = K * np.float32([w / nw, h / nh, 1]).reshape(3, 1)
        boxes.append(np.stack([xs.min(1), ys.min(1), xs.max(1), ys.max(1)], 1))
        p = v[:, 4:7] @ K.T
        vecs.append(v), pcs.append(p[:, :2] / p[:, 2:3])
        scores.append(o["scores"]), scores3d.append(o["scores_3d"]), classes.append(o["pred_classes"])
    boxes = torch.from_numpy(np.concatenate(boxes)).float()
    vecs = torch.from_numpy(np.concatenate(vecs)).float()
    pcs = torch.from_numpy(np.concatenate(pcs)).float()
    scores, scores3d, classes = torch.cat(scores), torch.cat(scores3d), torch.cat(classes)
    keep = torch.arange(len(boxes))
    if len(boxes) > 0:
        if inf["DO_NMS"]:
            keep = O.batched_nms(boxes, scores3d, classes, cfg["DD3D"]["FCOS2D"]["INFERENCE"]["NMS_THRESH"])
        if inf["DO_BEV_NMS"]:
            v = vecs[keep]
            glob = N.boxes_to_global(v, CAM_TO_VEHICLE_QUAT, (0.0, 0.0, 0.0))  # bev_nms: camera -> vehicle frame (no translation)
            rot = N.boxes3d_to_rotated_boxes_global(glob[:, :4], glob[:, 4:7], glob[:, 7:])
            k2 = N.batched_nms_rotated(rot, scores3d[keep], classes[keep], inf["BEV_NMS_IOU_THRESH"])
            keep = keep[k2]
    return dict(pred_boxes=boxes[keep], vec=vecs[keep], proj_ctr=pcs[keep], scores=scores[keep], scores_3d=scores3d[keep], pred_classes=classes[keep],
                n_union=len(boxes))


def _merge_one_image(sd, cfg, x, forward, extra_keys=()):
    """_inference_one_image / _get_augmented_instances for one image; `forward(sd, cfg, inputs) -> (results, stages)`.
    Returns the merged fields after NMS / BEV NMS as tensors."""
    inf, aug = cfg["DD3D"]["INFERENCE"], cfg["TEST"]["AUG"]
    copies = augment(x, aug["MIN_SIZES"], aug["MAX_SIZE"], aug["FLIP"])
    bs = cfg["TEST"]["IMS_PER_BATCH"]
    outputs = []
    for i in range(0, len(copies), bs):
        res, _ = forward(sd, cfg, [c[0] for c in copies[i:i + bs]])
        outputs += res
    boxes, vecs, pcs, K0 = [], [], [], None
    for (d, ((h, w, nh, nw), fw)), o in zip(copies, outputs):
        b = o["pred_boxes"].numpy().astype(np.float32)
        xs, ys = b[:, [0, 2, 0, 2]].copy(), b[:, [1, 1, 3, 3]].copy()
        v = O.boxes3d_vectorize(o["pred_boxes3d"]).numpy().astype(np.float32)
        K = d["intrinsics"].numpy().astype(np.float32).copy()
        if fw is not None:
            xs = fw - xs
            v = np.concatenate([v[:, [3]], -v[:, [2]], -v[:, [1]], v[:, [0]], -v[:, 4:5], v[:, 5:7], v[:, 7:]], axis=1)
            K[0, 2] = fw - K[0, 2]
        xs, ys = xs * (w * 1.0 / nw), ys * (h * 1.0 / nh)
        K = K * np.float32([w / nw, h / nh, 1]).reshape(3, 1)
        K0 = K if K0 is None else K0
        boxes.append(np.stack([xs.min(1), ys.min(1), xs.max(1), ys.max(1)], 1))
        p = v[:, 4:7] @ K.T
        vecs.append(v), pcs.append(p[:, :2] / p[:, 2:3])
    f = dict(pred_boxes=torch.from_numpy(np.concatenate(boxes)).float(), vec=torch.from_numpy(np.concatenate(vecs)).float(),
             proj_ctr=torch.from_numpy(np.concatenate(pcs)).float())
    for k in ("scores", "scores_3d", "pred_classes") + tuple(extra_keys):
        f[k] = torch.cat([o[k] for o in outputs])
    keep = torch.arange(len(f["scores"]))
    if len(keep) > 0:
        if inf["DO_NMS"]:
            keep = O.batched_nms(f["pred_boxes"], f["scores_3d"], f["pred_classes"], cfg["DD3D"]["FCOS2D"]["INFERENCE"]["NMS_THRESH"])
        if inf["DO_BEV_NMS"]:
            glob = N.boxes_to_global(f["vec"][keep], CAM_TO_VEHICLE_QUAT, (0.0, 0.0, 0.0))
            rot = N.boxes3d_to_rotated_boxes_global(glob[:, :4], glob[:, 4:7], glob[:, 7:])
            keep = keep[N.batched_nms_rotated(rot, f["scores_3d"][keep], f["pred_classes"][keep], inf["BEV_NMS_IOU_THRESH"])]
    return {k: v[keep] for k, v in f.items()}, K0


def nuscenes_tta_forward(sd, cfg, batched_inputs):
    """nuscenes_dd3d_tta.py:40-74: per-image TTA merge, then nuscenes_sample_aggregate over the merged instances."""
    merged = []
    for x in batched_inputs:
        f, K = _merge_one_image(sd, cfg, x, N.nuscenes_dd3d_forward, ("pred_attributes", "pred_speeds"))
        n = len(f["scores"])
        inv_K = torch.from_numpy(np.linalg.inv(K).astype(np.float32))
        f["pred_boxes3d"] = dict(quat=f["vec"][:, :4], proj_ctr=f["proj_ctr"], depth=f["vec"][:, 6:7], size=f["vec"][:, 7:10],
                                 inv_intrinsics=inv_K[None].expand(n, 3, 3))
        merged.append({k: v for k, v in f.items() if k not in ("vec", "proj_ctr")})
    nus = cfg["DD3D"]["NUSC"]["INFERENCE"]
    groups = N.get_group_idxs([x["sample_token"] for x in batched_inputs], nus["NUM_IMAGES_PER_SAMPLE"])
    poses = [N._pose_tuple(x["pose"]) for x in batched_inputs]
    out, _ = N.nuscenes_sample_aggregate(merged, groups, cfg["DD3D"]["NUM_CLASSES"], poses, cfg["DD3D"]["INFERENCE"]["BEV_NMS_IOU_THRESH"],
                                         max_num_dets_per_sample=nus["MAX_NUM_DETS_PER_SAMPLE"])
    return out, merged
