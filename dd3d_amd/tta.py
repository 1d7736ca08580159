"""Test-time augmentation wrapper: ``DatasetMapperTTA`` / ``DD3DWithTTA`` with the reference's names, constructor arguments and call
contract (tridet/modeling/dd3d/test_time_augmentation.py:23-260).

Per input image: ``len(MIN_SIZES) * (2 if FLIP else 1)`` augmented copies (shortest-edge resize -- the Pillow-exact device resize
of dd3d_amd.inputs -- and horizontal flip), forwarded in batches of ``TEST.IMS_PER_BATCH // world_size`` exactly like the reference
(mixed sizes in one padded batch included), the boxes mapped back to the original frame (2D boxes, 3D boxes, intrinsics:
flip_transform.py:7-62, resize_transform.py:13-21,76-77), then one class-aware NMS over the union ranked by ``scores_3d`` and, if
``model.do_bev_nms``, the BEV rotated NMS in the vehicle frame -- both by the path's own kernels (dd3d_nms_finalize,
dd3d_bev_nms_aggregate) on a packed buffer instead of torchvision / detectron2 ops.
"""
import copy
import ctypes as C

import numpy as np
import torch
from torch import nn

from dd3d_amd import hip
from dd3d_amd.inputs import DeviceResizer, shortest_edge_size
from dd3d_amd.modeling.dd3d import DD3D
from dd3d_amd.structures import Boxes, Boxes3D, Instances

__all__ = ["DatasetMapperTTA", "DD3DWithTTA", "NuscenesDD3DWithTTA"]

# CAMERA_TO_VEHICLE_ROTATION (tridet/layers/bev_nms.py:27-32) as the (w, x, y, z) quaternion of [[0,0,1],[-1,0,0],[0,-1,0]]
_CAM_TO_VEHICLE_QUAT = (0.5, -0.5, 0.5, -0.5)


class TTATransform:
    """What DatasetMapperTTA did to one copy: [pre-resize to the input shape] + shortest-edge resize + optional horizontal flip;
    only the inverse maps are needed afterwards (fvcore TransformList.inverse of `pre_tfm + tfms`)."""
    def __init__(self, pre, resize, flip_width):
        self.pre, self.resize, self.flip_width = pre, resize, flip_width  # (h, w, new_h, new_w) tuples or None; width or None

    def apply_intrinsics(self, K):
        """Forward map of the augmentation on a (3, 3) float32 intrinsics matrix (resize, then flip)."""
        K = np.array(K, dtype=np.float32)
        h, w, nh, nw = self.resize
        K = K * np.float32([nw / w, nh / h, 1]).reshape(3, 1)  # apply_imresize_intrinsics
        if self.flip_width is not None:
            K[0, 2] = self.flip_width - K[0, 2]  # apply_hflip_intrinsics
        return K

    def inverse_intrinsics(self, K):
        K = np.array(K, dtype=np.float32)
        if self.flip_width is not None:
            K[0, 2] = self.flip_width - K[0, 2]
        for t in (self.resize, self.pre):
            if t is not None:
                h, w, nh, nw = t
                K = K * np.float32([w / nw, h / nh, 1]).reshape(3, 1)  # the inverse ResizeTransform maps (nh, nw) -> (h, w)
        return K

    def inverse_box(self, boxes):
        """(n, 4) float32 XYXY in the augmented frame -> original frame (corner-wise, then min / max, as Transform.apply_box)."""
        b = np.array(boxes, dtype=np.float32).reshape(-1, 4)
        x = b[:, [0, 2, 0, 2]].copy()
        y = b[:, [1, 1, 3, 3]].copy()
        if self.flip_width is not None:
            x = self.flip_width - x
        for t in (self.resize, self.pre):
            if t is not None:
                h, w, nh, nw = t
                x = x * (w * 1.0 / nw)
                y = y * (h * 1.0 / nh)
        return np.stack([x.min(1), y.min(1), x.max(1), y.max(1)], axis=1)

    def inverse_box3d(self, vec):
        """(n, 10) [quat wxyz, tvec, size]: the flip mirrors the box (apply_hflip_box3d); resizes leave it alone."""
        v = np.array(vec, dtype=np.float32).reshape(-1, 10)
        if self.flip_width is not None:
            v = np.concatenate([v[:, [3]], -v[:, [2]], -v[:, [1]], v[:, [0]], -v[:, 4:5], v[:, 5:7], v[:, 7:]], axis=1)
        return v


class DatasetMapperTTA:
    """test_time_augmentation.py:23-86.  Images stay on the device: resize by dd3d_resize_bilinear_u8 (bit-identical to PIL), flip by
    a device-side index flip."""
    def __init__(self, cfg, device="cuda"):
        self.min_sizes = list(cfg.TEST.AUG.MIN_SIZES)
        self.max_size = cfg.TEST.AUG.MAX_SIZE
        self.flip = cfg.TEST.AUG.FLIP
        self.image_format = cfg.INPUT.FORMAT
        self.resizer = DeviceResizer(device)

    def __call__(self, dataset_dict):
        image = dataset_dict["image"].to(self.resizer.device, non_blocking=True)
        _, h, w = image.shape
        orig_shape = (dataset_dict["height"], dataset_dict["width"])
        pre = None if (h, w) == tuple(orig_shape) else (orig_shape[0], orig_shape[1], h, w)
        rest = {k: v for k, v in dataset_dict.items() if k != "image"}
        ret = []
        for min_size in self.min_sizes:
            nh, nw = shortest_edge_size(h, w, min_size, self.max_size)
            resized = self.resizer(image, nh, nw)
            for flip in ((False, True) if self.flip else (False, )):
                dic = copy.deepcopy(rest)
                tfm = TTATransform(pre, (h, w, nh, nw), nw if flip else None)
                dic["transforms"] = tfm
                dic["image"] = torch.flip(resized, dims=[2]) if flip else resized
                if "intrinsics" in dic:
                    K = tfm.apply_intrinsics(dic["intrinsics"].cpu().numpy().astype(np.float32))
                    dic["intrinsics"] = torch.as_tensor(K)
                    dic["inv_intrinsics"] = torch.as_tensor(np.linalg.inv(K))
                ret.append(dic)
        return ret


class _MergeNMS:
    """Class-aware NMS (+ BEV rotated NMS) over a union of detections with the forward path's kernels."""
    def __init__(self, device):
        self.device = device

    def __call__(self, boxes, boxes3d_vec, proj_ctr, depth, scores, scores_3d, classes, inv_K, do_nms, nms_thresh, do_bev, bev_thresh, num_classes,
                 attributes=None, speeds=None):
        dev, n = self.device, boxes.shape[0]
        if n > 8192:
            raise NotImplementedError(f"TTA merge of {n} boxes exceeds the 8192-box sorter")
        lib = hip.lib()
        cand = torch.zeros((1, hip.CAND_FIELDS, n), dtype=torch.float32, device=dev)
        cand[0, 0:4] = boxes.T
        cand[0, 4], cand[0, 5] = scores, scores_3d
        cand[0, 6] = classes.to(torch.int32).view(torch.float32)
        cand[0, 7] = torch.arange(n, dtype=torch.int32, device=dev).view(torch.float32)
        cand[0, 10:14], cand[0, 14:16], cand[0, 16], cand[0, 17:20] = boxes3d_vec[:, 0:4].T, proj_ctr.T, depth, boxes3d_vec[:, 7:10].T
        if attributes is not None:
            cand[0, 20], cand[0, 21] = attributes.to(torch.int32).view(torch.float32), speeds
        counts = torch.tensor([[n]], dtype=torch.int32, device=dev)
        ncap = (n + 63) // 64 * 64
        a = hip.NmsArgs()
        keep = dict(sort_idx=torch.zeros((1, ncap), dtype=torch.int32, device=dev), sbox=torch.zeros((1, ncap, 4), dtype=torch.float32, device=dev),
                    scls=torch.zeros((1, ncap), dtype=torch.int32, device=dev), mask=torch.zeros((1, ncap, ncap // 64), dtype=torch.int64, device=dev),
                    nvalid=torch.zeros((1, 2), dtype=torch.int32, device=dev), det=torch.zeros((1, n, hip.DET_FIELDS), dtype=torch.float32, device=dev),
                    det_count=torch.zeros((1, ), dtype=torch.int32, device=dev), out_size=torch.ones((1, 4), dtype=torch.float32, device=dev))
        a.cand, a.counts, a.G, a.num_levels, a.topk = cand.data_ptr(), counts.data_ptr(), 1, 1, n
        a.do_nms, a.use_score3d, a.nms_thresh, a.post_topk, a.do_postprocess = int(bool(do_nms)), 1, float(nms_thresh), 0, 0
        a.out_size, a.sort_idx, a.sbox, a.scls = keep["out_size"].data_ptr(), keep["sort_idx"].data_ptr(), keep["sbox"].data_ptr(), keep["scls"].data_ptr()
        a.mask, a.nvalid, a.det, a.det_count, a.det_cap = keep["mask"].data_ptr(), keep["nvalid"].data_ptr(), keep["det"].data_ptr(), keep["det_count"].data_ptr(), n
        st = hip.current_stream()
        hip.check(lib.dd3d_nms_finalize(C.byref(a), st), "tta nms")
        det, det_count = keep["det"], keep["det_count"]
        if do_bev:
            b = hip.BevArgs()
            pose = torch.tensor([list(_CAM_TO_VEHICLE_QUAT) + [0.0, 0.0, 0.0]], dtype=torch.float32, device=dev)
            group = torch.zeros((1, ), dtype=torch.int32, device=dev)
            ncapb = ncap
            work = dict(work=torch.zeros((n, 16), dtype=torch.float32, device=dev), sbox=torch.zeros((n, 8), dtype=torch.float32, device=dev),
                        mask=torch.zeros((min(ncapb, 8192), min(ncapb, 8192) // 64), dtype=torch.int64, device=dev), meta=torch.zeros((4, ), dtype=torch.int32, device=dev),
                        det_out=torch.zeros_like(det), count_out=torch.zeros_like(det_count), invK=inv_K.reshape(1, 9).contiguous().to(dev))
            b.det_in, b.count_in, b.inv_K, b.pose, b.group = det.data_ptr(), det_count.data_ptr(), work["invK"].data_ptr(), pose.data_ptr(), group.data_ptr()
            b.out_size, b.G, b.det_cap, b.num_classes = keep["out_size"].data_ptr(), 1, n, int(num_classes)
            b.iou_thresh, b.max_dets, b.write_global, b.do_postprocess = float(bev_thresh), 0, 0, 0
            b.work, b.sbox, b.mask, b.meta = work["work"].data_ptr(), work["sbox"].data_ptr(), work["mask"].data_ptr(), work["meta"].data_ptr()
            b.det_out, b.count_out = work["det_out"].data_ptr(), work["count_out"].data_ptr()
            hip.check(lib.dd3d_bev_nms_aggregate(C.byref(b), st), "tta bev nms")
            # the reference's bev_nms returns its keep list in descending score order and indexes with it; the aggregate kernel
            # keeps the input (NMS output = score-descending) order, which is the same thing
            det, det_count = work["det_out"], work["count_out"]
        k = int(det_count.cpu()[0])
        return det[0, :k]


class DD3DWithTTA(nn.Module):
    """test_time_augmentation.py:89-260."""
    def __init__(self, cfg, model, tta_mapper=None):
        super().__init__()
        if isinstance(model, nn.parallel.DistributedDataParallel):
            model = model.module
        assert isinstance(model, DD3D), "DD3DwithTTA only supports on DD3D. Got a model of type {}".format(type(model))
        assert not model.postprocess_in_inference, "To use test-time augmentation, `postprocess_in_inference` must be False."
        self.cfg = copy.deepcopy(cfg)
        self.model = model
        self.nms_thresh = cfg.DD3D.FCOS2D.INFERENCE.NMS_THRESH
        self.tta_mapper = tta_mapper if tta_mapper is not None else DatasetMapperTTA(cfg, model.device)
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        self.batch_size = cfg.TEST.IMS_PER_BATCH // world
        self._merge = _MergeNMS(model.device)

    def _batch_inference(self, batched_inputs):
        outputs, inputs = [], []
        for idx, inp in enumerate(batched_inputs):
            inputs.append(inp)
            if len(inputs) == self.batch_size or idx == len(batched_inputs) - 1:
                # NMS per augmented image happens inside the model; results are copied out of the plan's buffers before the next batch
                outputs.extend([res["instances"] for res in self.model(inputs)])
                inputs = []
        return outputs

    def __call__(self, batched_inputs):
        def _with_size(d):
            ret = copy.copy(d)
            if "image" not in ret:
                raise NotImplementedError("reading images from file_name is the data pipeline's job (detectron2 read_image)")
            if "height" not in ret and "width" not in ret:
                ret["height"], ret["width"] = ret["image"].shape[1], ret["image"].shape[2]
            return ret

        return [self._inference_one_image(_with_size(x)) for x in batched_inputs]

    @torch.no_grad()
    def _merged_detections(self, x, with_nusc=False):
        """Union of the augmented forwards mapped back to the original frame, after the merge NMS (+ BEV NMS): (det [k][DET_FIELDS] in
        the layout of dd3d_nms_finalize, inverse original intrinsics (3, 3) on the host)."""
        augmented_inputs = self.tta_mapper(x)
        tfms = [a.pop("transforms") for a in augmented_inputs]
        outputs = self._batch_inference(augmented_inputs)
        dev = self.model.device
        boxes, vecs, scores, scores_3d, classes, attrs, speeds = [], [], [], [], [], [], []
        orig_K = None
        for inp, out, tfm in zip(augmented_inputs, outputs, tfms):
            boxes.append(tfm.inverse_box(out.pred_boxes.tensor.cpu().numpy()))
            vecs.append(tfm.inverse_box3d(out.pred_boxes3d.vectorize().cpu().numpy()))
            K = tfm.inverse_intrinsics(inp["intrinsics"].cpu().numpy())
            orig_K = K if orig_K is None else orig_K
            # Boxes3D.from_vectors(vecs, orig_intrinsics): proj_ctr = K t / z, depth = z   (boxes3d.py:176-217), per augmented copy
            p = vecs[-1][:, 4:7] @ K.T
            vecs[-1] = np.concatenate([vecs[-1], p[:, :2] / p[:, 2:3]], axis=1)  # columns 10, 11: proj_ctr
            scores.append(out.scores), scores_3d.append(out.scores_3d), classes.append(out.pred_classes)
            if with_nusc:
                attrs.append(out.pred_attributes), speeds.append(out.pred_speeds)
        boxes = torch.from_numpy(np.concatenate(boxes, 0)).to(dev)
        vecs = torch.from_numpy(np.concatenate(vecs, 0).astype(np.float32)).to(dev)
        scores, scores_3d, classes = torch.cat(scores), torch.cat(scores_3d), torch.cat(classes)
        attrs = torch.cat(attrs) if with_nusc else torch.zeros_like(classes)
        speeds = torch.cat(speeds) if with_nusc else torch.zeros_like(scores)
        n = boxes.shape[0]
        bev = (not self.model.only_box2d) and bool(self.model.do_bev_nms)
        if bev and not self.model.do_nms and n > 0:
            # bev_nms alone returns its keep list in descending score order; the aggregate kernel preserves input order
            order = torch.argsort(scores_3d, descending=True, stable=True)
            boxes, vecs, scores, scores_3d, classes = boxes[order], vecs[order], scores[order], scores_3d[order], classes[order]
            attrs, speeds = attrs[order], speeds[order]
        inv_K = torch.from_numpy(np.linalg.inv(orig_K).astype(np.float32))
        if n == 0 or not (self.model.do_nms or bev):
            det = torch.zeros((n, hip.DET_FIELDS), dtype=torch.float32, device=dev)
            if n:
                det[:, 0:4], det[:, 4], det[:, 5], det[:, 6] = boxes, scores, scores_3d, classes.float()
                det[:, 10:14], det[:, 14:16], det[:, 16], det[:, 17:20] = vecs[:, 0:4], vecs[:, 10:12], vecs[:, 6], vecs[:, 7:10]
                det[:, 20], det[:, 21] = attrs.float(), speeds
            return det, inv_K
        det = self._merge(boxes, vecs, vecs[:, 10:12], vecs[:, 6], scores, scores_3d, classes, inv_K, self.model.do_nms, self.nms_thresh, bev,
                          self.model.bev_nms_iou_thresh, self.model.num_classes, attrs, speeds)
        return det, inv_K

    def _instances(self, det, inv_K, image_size, with_nusc=False, with_global=False):
        dev, k = self.model.device, det.shape[0]
        r = Instances(image_size)
        r.pred_boxes = Boxes(det[:, 0:4].contiguous())
        r.pred_boxes3d = Boxes3D(det[:, 10:14].contiguous(), det[:, 14:16].contiguous(), det[:, 16:17].contiguous(), det[:, 17:20].contiguous(),
                                 inv_K.to(dev)[None].expand(k, 3, 3))
        r.pred_classes = det[:, 6].to(torch.int64)
        r.scores, r.scores_3d = det[:, 4].contiguous(), det[:, 5].contiguous()
        if with_nusc:
            r.pred_attributes, r.pred_speeds = det[:, 20].to(torch.int64), det[:, 21].contiguous()
        if with_global:
            from dd3d_amd.structures import GenericBoxes3D
            r.pred_boxes3d_global = GenericBoxes3D(det[:, 22:26].contiguous(), det[:, 26:29].contiguous(), det[:, 17:20].contiguous())
        return r

    def _inference_one_image(self, x):
        det, inv_K = self._merged_detections(x)
        return {"instances": self._instances(det, inv_K, (x["height"], x["width"]))}


class NuscenesDD3DWithTTA(DD3DWithTTA):
    """nuscenes_dd3d_tta.py:21-178: DD3DWithTTA per camera image (attributes / speeds ride through the merge), then the cross-camera
    sample aggregation (postprocessing.py:58-108) over the merged per-image detections."""
    def __init__(self, cfg, model, tta_mapper=None):
        from dd3d_amd.modeling.nuscenes_dd3d import NuscenesDD3D
        assert isinstance(model, NuscenesDD3D), \
            "NuscenesDD3DWithTTA only supports on NuscenesDD3D. Got a model of type {}".format(type(model))
        super().__init__(cfg, model, tta_mapper)

    @torch.no_grad()
    def __call__(self, batched_inputs):
        from dd3d_amd.modeling.nuscenes_dd3d import get_group_idxs
        xs = []
        for d in batched_inputs:
            r = copy.copy(d)
            if "image" not in r:
                raise NotImplementedError("reading images from file_name is the data pipeline's job (detectron2 read_image)")
            if "height" not in r and "width" not in r:
                r["height"], r["width"] = r["image"].shape[1], r["image"].shape[2]
            xs.append(r)
        merged = [self._merged_detections(x, with_nusc=True) for x in xs]
        G, dev = len(xs), self.model.device
        groups = get_group_idxs([x["sample_token"] for x in batched_inputs], self.model.num_images_per_sample)
        group_of = [0] * G
        for gi, idxs in enumerate(groups.values()):
            for i in idxs:
                group_of[i] = gi
        cap = max(1, max(d.shape[0] for d, _ in merged))
        if G * cap > 8192:
            raise NotImplementedError(f"sample aggregation over {G} x {cap} boxes exceeds the 8192-box sorter")
        det_in = torch.zeros((G, cap, hip.DET_FIELDS), dtype=torch.float32, device=dev)
        for i, (d, _) in enumerate(merged):
            det_in[i, :d.shape[0]] = d
        count_in = torch.tensor([d.shape[0] for d, _ in merged], dtype=torch.int32, device=dev)
        inv_K = torch.stack([k for _, k in merged]).reshape(G, 9).to(dev)
        pose = torch.tensor([[float(v) for v in x["pose"].quat.elements] + [float(v) for v in x["pose"].tvec] for x in batched_inputs],
                            dtype=torch.float32, device=dev)
        group = torch.tensor(group_of, dtype=torch.int32, device=dev)
        ncap = (G * cap + 63) // 64 * 64
        work = dict(work=torch.zeros((G * cap, 16), dtype=torch.float32, device=dev), sbox=torch.zeros((G * cap, 8), dtype=torch.float32, device=dev),
                    mask=torch.zeros((min(ncap, 8192), min(ncap, 8192) // 64), dtype=torch.int64, device=dev), meta=torch.zeros((4, ), dtype=torch.int32, device=dev),
                    det_out=torch.zeros_like(det_in), count_out=torch.zeros_like(count_in), out_size=torch.ones((G, 4), dtype=torch.float32, device=dev))
        b = hip.BevArgs()
        b.det_in, b.count_in, b.inv_K, b.pose, b.group = det_in.data_ptr(), count_in.data_ptr(), inv_K.data_ptr(), pose.data_ptr(), group.data_ptr()
        b.out_size, b.G, b.det_cap, b.num_classes = work["out_size"].data_ptr(), G, cap, int(self.model.num_classes)
        b.iou_thresh, b.max_dets = float(self.model.bev_nms_iou_thresh), int(self.model.max_num_dets_per_sample)
        b.write_global, b.do_postprocess = 1, 0
        b.work, b.sbox, b.mask, b.meta = work["work"].data_ptr(), work["sbox"].data_ptr(), work["mask"].data_ptr(), work["meta"].data_ptr()
        b.det_out, b.count_out = work["det_out"].data_ptr(), work["count_out"].data_ptr()
        hip.check(hip.lib().dd3d_bev_nms_aggregate(C.byref(b), hip.current_stream()), "tta sample aggregate")
        counts = work["count_out"].cpu().tolist()
        out = []
        for i, x in enumerate(xs):
            d = work["det_out"][i, :counts[i]]
            out.append({"instances": self._instances(d, merged[i][1], (x["height"], x["width"]), with_nusc=True, with_global=True)})
        return out
