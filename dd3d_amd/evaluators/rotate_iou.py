"""Drop-in for the overlap functions of tridet/evaluators/rotate_iou.py (the reference's numba.cuda kernels do not exist on
ROCm): same names, numpy in / numpy out, same argument meaning -- `rotate_iou_gpu_eval(boxes, query_boxes, criterion, device_id)`
(:292-327), `d3_box_overlap_kernel(boxes, qboxes, rinc, criterion, camera_coordinate)` (:330-357, in place on `rinc`),
`image_box_overlap(boxes, query_boxes, criterion)` (:360-381) -- running on the MI355X through libdd3d_hip.so.
`kitti_3d_evaluator.py:622-632` (bev_box_overlap / d3_box_overlap) works unchanged on top of them.
"""
import numpy as np
import torch

from dd3d_amd import hip


def _dev(a, cols, device_id):
    t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols))
    return t.to(torch.device("cuda", device_id))


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=boxes.dtype)
    b, q = _dev(boxes, 5, device_id), _dev(query_boxes, 5, device_id)
    out = torch.empty((N, K), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        hip.check(hip.lib().dd3d_rotate_iou_eval(b.data_ptr(), q.data_ptr(), out.data_ptr(), N, K, int(criterion), hip.current_stream()),
                  "rotate_iou_eval")
    return out.cpu().numpy().astype(boxes.dtype)


def d3_box_overlap_kernel(boxes, qboxes, rinc, criterion=-1, camera_coordinate=False, device_id=0):
    """In place on `rinc`, like the reference's numba kernel."""
    N, K = boxes.shape[0], qboxes.shape[0]
    if N == 0 or K == 0:
        return
    b, q = _dev(boxes, 7, device_id), _dev(qboxes, 7, device_id)
    r = torch.as_tensor(np.ascontiguousarray(rinc, dtype=np.float32)).to(b.device)
    with torch.cuda.device(b.device):
        hip.check(hip.lib().dd3d_d3_box_overlap(b.data_ptr(), q.data_ptr(), r.data_ptr(), N, K, int(criterion), int(bool(camera_coordinate)),
                                                hip.current_stream()), "d3_box_overlap")
    rinc[...] = r.cpu().numpy().astype(rinc.dtype)


def d3_box_overlap(boxes, qboxes, criterion=-1, camera_coordinate=True, device_id=0):
    """KITTI3DEvaluator.d3_box_overlap (kitti_3d_evaluator.py:628-632): BEV intersection areas, then the vertical overlap."""
    rinc = rotate_iou_gpu_eval(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2, device_id)
    d3_box_overlap_kernel(boxes, qboxes, rinc, criterion, camera_coordinate, device_id)
    return rinc


def image_box_overlap(boxes, query_boxes, criterion=-1, device_id=0):
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=boxes.dtype)
    b, q = _dev(boxes, 4, device_id), _dev(query_boxes, 4, device_id)
    out = torch.empty((N, K), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        hip.check(hip.lib().dd3d_image_box_overlap(b.data_ptr(), q.data_ptr(), out.data_ptr(), N, K, int(criterion), hip.current_stream()),
                  "image_box_overlap")
    return out.cpu().numpy().astype(boxes.dtype)
