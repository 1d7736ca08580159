"""Batched result formatting on the MI355X: the arithmetic of
tridet/evaluators/kitti_3d_evaluator.py:205-264 (`convert_3d_box_to_kitti`) and nuscenes_evaluator.py:196-198 (global velocity)
for every detection of a batch in one launch of `dd3d_format_boxes3d` and one device->host copy.  The reference does this box by
box on the host (pyquaternion objects, several `.cpu()` copies per detection).
"""
import numpy as np
import torch

from dd3d_amd import hip

FIELDS = ("W", "L", "H", "x", "y", "z", "rot_y", "alpha", "vx", "vy")


def _device_of(t):
    return t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())


def format_boxes3d(box3d_vec, quat_global=None, speeds=None):
    """box3d_vec (n,10) float32 = `Boxes3D.vectorize()`; optional quat_global (n,4) + speeds (n,) -> numpy float64 (n,10) with the
    columns of `FIELDS`.  Host tensors are uploaded; there is no CPU implementation."""
    box3d_vec = torch.as_tensor(box3d_vec, dtype=torch.float32).reshape(-1, 10)
    n = box3d_vec.shape[0]
    if n == 0:
        return np.zeros((0, 10), dtype=np.float64)
    hip.lib()  # HipLibraryMissing when the extension is not built
    dev = _device_of(box3d_vec)
    v = box3d_vec.to(dev).contiguous()
    if (quat_global is None) != (speeds is None):
        raise ValueError("quat_global and speeds go together")
    q = s = None
    if quat_global is not None:
        q = torch.as_tensor(quat_global, dtype=torch.float32).reshape(-1, 4).to(dev).contiguous()
        s = torch.as_tensor(speeds, dtype=torch.float32).reshape(-1).to(dev).contiguous()
        if q.shape[0] != n or s.shape[0] != n:
            raise ValueError(f"quat_global / speeds must have {n} rows")
    out = torch.empty((n, 10), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        hip.check(hip.lib().dd3d_format_boxes3d(v.data_ptr(), q.data_ptr() if q is not None else None,
                                                s.data_ptr() if s is not None else None, out.data_ptr(), n, hip.current_stream()),
                  "format_boxes3d")
    return out.cpu().numpy()


def kitti_tuple(row):
    """One row of `format_boxes3d` -> the 8-tuple `convert_3d_box_to_kitti` returns, with the reference's scalar types (sizes and
    translation float32, rot_y a Python float, alpha numpy float64) so that DataFrame / csv output is character-identical."""
    return (np.float32(row[0]), np.float32(row[1]), np.float32(row[2]), np.float32(row[3]), np.float32(row[4]), np.float32(row[5]),
            float(row[6]), np.float64(row[7]))


def xyxy_to_xywh(box):
    """detectron2 BoxMode.convert(list, XYXY_ABS, XYWH_ABS) [ext]: float64 numpy arithmetic on the listed values, list out."""
    a = np.array(box, dtype=np.float64)
    a[2] -= a[0]
    a[3] -= a[1]
    return a.tolist()
