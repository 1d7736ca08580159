"""Golden vectors for the evaluator-side overlap kernels, produced by the REFERENCE'S OWN device functions
(tridet/evaluators/rotate_iou.py: devRotateIoUEval and everything it calls, d3_box_overlap_kernel, image_box_overlap), executed
as plain Python: numba / mpi4py are not installed, so `numba.cuda.jit` & co. are replaced by identity decorators and
`cuda.local.array` by numpy float32 arrays (every arithmetic step then runs in float32 exactly as written; only math.cos / sin /
sqrt are evaluated in double and rounded once, an O(1e-7) difference from the compiled kernels).

    python tests/golden/make_rotate_iou_golden.py        ->  tests/golden/rotate_iou.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("DD3D_REFERENCE_ROOT", "/root/reference")


def _install_shims():
    import torch

    def _identity_decorator(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba = types.ModuleType("numba")
    numba.jit = _identity_decorator
    numba.float32 = np.float32
    cuda = types.ModuleType("numba.cuda")
    cuda.jit = _identity_decorator
    cuda.select_device = lambda *_: None
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype=np.float32: np.zeros(shape, dtype=np.float32))
    cuda.shared = types.SimpleNamespace(array=lambda shape, dtype=np.float32: np.zeros(shape, dtype=np.float32))
    errors = types.ModuleType("numba.errors")
    errors.NumbaDeprecationWarning = DeprecationWarning
    numba.cuda, numba.errors = cuda, errors
    sys.modules.update({"numba": numba, "numba.cuda": cuda, "numba.errors": errors})
    mpi = types.ModuleType("mpi4py")
    mpi.MPI = types.SimpleNamespace(COMM_WORLD=types.SimpleNamespace(Get_rank=lambda: 0))
    sys.modules["mpi4py"] = mpi
    torch.cuda.device_count = lambda: 1  # module-level `rank % device_count()` of the reference file
    for pkg in ("tridet", "tridet.evaluators"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def make_boxes(rng, n, spread=6.0):
    """(x, y, x_d, y_d, angle) BEV boxes clustered enough to overlap often, with a few special cases appended by the caller."""
    ctr = rng.uniform(-spread, spread, (n, 2))
    dims = rng.uniform(0.6, 4.5, (n, 2))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([ctr, dims, ang], 1).astype(np.float32)


def main():
    _install_shims()
    from tridet.evaluators import rotate_iou as R  # the reference's own code
    rng = np.random.default_rng(7)
    boxes = make_boxes(rng, 40)
    q = make_boxes(rng, 28)
    # special cases: identical, same centre rotated, contained, edge-touching axis-aligned, far apart, quarter-turn
    boxes[0] = [0, 0, 2, 1, 0.3]
    q[0] = boxes[0]
    q[1] = [0, 0, 2, 1, 0.3 + np.pi / 2]
    boxes[1] = [1, 1, 4, 4, 0.0]
    q[2] = [1, 1, 1, 1, 0.7]
    boxes[2] = [0, 0, 2, 2, 0.0]
    q[3] = [2, 0, 2, 2, 0.0]
    q[4] = [50, 50, 1, 1, 0.1]
    out = {"boxes": boxes, "qboxes": q}
    for crit in (-1, 0, 1, 2):
        iou = np.zeros((len(boxes), len(q)), dtype=np.float32)
        for i in range(len(boxes)):
            for j in range(len(q)):
                # rotate_iou_kernel_eval (rotate_iou.py:260-289): dev_iou[i, j] = devRotateIoUEval(query_box j, box i, criterion)
                iou[i, j] = R.devRotateIoUEval(q[j].copy(), boxes[i].copy(), crit)
        out[f"riou_{crit}"] = iou
    # 3D overlap in camera coordinates (kitti_3d_evaluator.py:628-632): boxes = (x, y, z, l, h, w, ry)
    b3 = np.concatenate([boxes[:, :1], rng.uniform(0.5, 2.5, (len(boxes), 1)), boxes[:, 1:2], boxes[:, 2:3], rng.uniform(1.0, 2.5, (len(boxes), 1)),
                         boxes[:, 3:4], boxes[:, 4:5]], 1).astype(np.float32)
    q3 = np.concatenate([q[:, :1], rng.uniform(0.5, 2.5, (len(q), 1)), q[:, 1:2], q[:, 2:3], rng.uniform(1.0, 2.5, (len(q), 1)), q[:, 3:4], q[:, 4:5]],
                        1).astype(np.float32)
    out["boxes3d"], out["qboxes3d"] = b3, q3
    for crit in (-1, 0, 1):
        for cam in (True, False):
            rinc = out["riou_2"].copy()  # rotate_iou_gpu_eval(boxes[:, [0, 2, 3, 5, 6]], ..., 2) of the same BEV boxes
            R.d3_box_overlap_kernel(b3, q3, rinc, crit, cam)
            out[f"d3_{crit}_{int(cam)}"] = rinc
    # 2D image boxes (image_box_overlap :360-381)
    xy = rng.uniform(0, 100, (30, 2))
    ib = np.concatenate([xy, xy + rng.uniform(1, 40, (30, 2))], 1).astype(np.float32)
    xy = rng.uniform(0, 100, (22, 2))
    iq = np.concatenate([xy, xy + rng.uniform(1, 40, (22, 2))], 1).astype(np.float32)
    out["iboxes"], out["iqboxes"] = ib, iq
    for crit in (-1, 0, 1):
        out[f"image_{crit}"] = R.image_box_overlap(ib, iq, crit)
    path = os.path.join(HERE, "rotate_iou.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith(("riou", "d3"))}, "mean iou", float(out["riou_-1"].mean()),
          "nonzero", int((out["riou_-1"] > 0).sum()))


if __name__ == "__main__":
    main()
