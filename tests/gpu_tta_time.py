"""Dev tool (GPU): timing lines of the rows SURVEY section 8(f) marks "next" -- they have parity tests, this gives them a number.

  f1  DD3DWithTTA at the experiment's own scales (configs/experiments/dd3d_kitti_dla34.yaml:44-53: MIN_SIZES [320 ... 576] x flip on a raw
      370 x 1224 frame): the reference's batching (ten copies in ONE batch of IMS_PER_BATCH on a 640 x 1920 canvas) and the per-scale
      batching (IMS_PER_BATCH 2: a scale's copy and its mirror share a launch plan on the scale's own canvas -- five plans), launch plans
      pre-built, hipGraph replay;
  f2  rotate_iou_gpu_eval / d3_box_overlap at the size KITTI3DEvaluator calls them with (kitti_3d_evaluator.py calculate_iou_partly: the
      boxes of one part of the validation set = 3769 images / 100 parts; ~8 ground-truth and ~30 detected boxes per image);
  f3  DeviceInputMapper: raw 370 x 1224 uint8 frame -> shortest-edge-384 resize (Pillow-exact) + intrinsics, on the device.

    python tests/gpu_tta_time.py > gpurun_out/r06_tta.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_state_dict  # noqa: E402
from tests.golden.make_tta_golden import FULL_TTA_OVERRIDES, full_tta_case  # noqa: E402


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def tta_lines():
    from dd3d_amd.structures import Pose
    from dd3d_amd.tta import DD3DWithTTA
    for label, bs in (("reference batching: 10 copies, one 640x1920 plan (IMS_PER_BATCH 80)", 80), ("per-scale batching: 5 plans of 2 copies (IMS_PER_BATCH 2)", 2)):
        cfg = get_cfg("dd3d_kitti_dla34", dict(FULL_TTA_OVERRIDES, TEST={"IMS_PER_BATCH": bs}))
        model = build_model(cfg)
        model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
        model.use_graph = True
        model.max_cached_plans = 8
        tta = DD3DWithTTA(cfg, model)
        x = full_tta_case()
        x["extrinsics"] = Pose()
        x["image"] = x["image"].cuda()  # resident, like the bench's inputs
        n = len(tta([x])[0]["instances"])
        ms = timed(lambda: tta([x]), reps=10)
        plans = sorted((p.B, p.Hp, p.Wp) for p in model._plans.values())
        gflop = sum(220.77 * p.B * p.Hp * p.Wp / (384.0 * 1280.0) for p in model._plans.values())
        print(f"f1 TTA  {label}: {ms:8.2f} ms per original image = {1e3 / ms:6.1f} img/s ({10e3 / ms:7.1f} augmented forwards/s, "
              f"{gflop / ms:6.1f} TFLOP/s f32-equivalent on the padded canvases); {n} merged detections; plans {plans}")


def nusc_tta_line():
    """NuscenesDD3DWithTTA at the nuScenes experiment's own scales (configs/experiments/dd3d_nusc_dla34.yaml:55-62: MIN_SIZES [640 ... 1152] x
    flip on one 6-camera sample of raw 900 x 1600 frames): per camera ten copies in ONE launch plan on the 1152 x 2048 canvas, then the
    sample aggregation (tests/golden/tta_nusc_dla34_scales.npz is the reference's result for the same sample)."""
    from dd3d_amd.tta import NuscenesDD3DWithTTA
    from tests.golden.make_tta_golden import NUSC_FULL_TTA_OVERRIDES, nusc_full_tta_case
    cfg = get_cfg("dd3d_nusc_dla34", NUSC_FULL_TTA_OVERRIDES)
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_nusc")))
    model.use_graph = True
    tta = NuscenesDD3DWithTTA(cfg, model)
    xs = nusc_full_tta_case()
    for x in xs:
        x["image"] = x["image"].cuda()
    n = sum(len(r["instances"]) for r in tta(xs))
    ms = timed(lambda: tta(xs), reps=5, warm=1)
    plans = sorted((p.B, p.Hp, p.Wp) for p in model._plans.values())
    gflop = 6 * sum(220.77 * p.B * p.Hp * p.Wp / (384.0 * 1280.0) for p in model._plans.values())
    print(f"f1 nuScenes TTA  one 6-camera sample, 5 scales x flip (60 augmented forwards on {plans}): {ms:8.2f} ms per sample = {6e3 / ms:6.1f} camera img/s "
          f"({60e3 / ms:7.1f} augmented forwards/s, {gflop / ms:6.1f} TFLOP/s f32-equivalent on the padded canvases); {n} detections after the sample aggregation")


def eval_lines():
    from dd3d_amd.evaluators import rotate_iou as R
    rng = np.random.default_rng(0)

    def boxes7(n):  # camera-frame (x, y, z, l, h, w, ry) boxes scattered over a KITTI-sized scene
        return np.concatenate([rng.uniform(-30, 30, (n, 1)), rng.uniform(0.5, 2.5, (n, 1)), rng.uniform(3, 70, (n, 1)),
                               rng.uniform(1.5, 4.5, (n, 1)), rng.uniform(1.3, 2.0, (n, 1)), rng.uniform(1.4, 2.0, (n, 1)),
                               rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)

    per_part = 38  # 3769 validation images / 100 parts
    gt, dt = boxes7(per_part * 8), boxes7(per_part * 30)
    bev = lambda b: b[:, [0, 2, 3, 5, 6]]
    for name, fn in (("rotate_iou_gpu_eval (BEV IoU)", lambda: R.rotate_iou_gpu_eval(bev(gt), bev(dt))),
                     ("d3_box_overlap (3D IoU)", lambda: R.d3_box_overlap(gt, dt))):
        ms = timed(fn, reps=20)
        print(f"f2 {name}: {gt.shape[0]} x {dt.shape[0]} boxes (one of 100 parts of KITTI val): {ms:7.3f} ms per call incl. H2D / D2H "
              f"= {100 * ms / 1e3:5.2f} s for the validation set ({gt.shape[0] * dt.shape[0] / ms / 1e3:7.1f} M pairs/s)")


def mapper_line():
    from dd3d_amd.inputs import DeviceInputMapper
    from dd3d_amd.synthetic import KITTI_K
    cfg = get_cfg("dd3d_kitti_dla34")
    mapper = DeviceInputMapper(cfg, "cuda")
    raw = torch.randint(0, 256, (3, 370, 1224), dtype=torch.uint8)
    K = torch.tensor(KITTI_K)
    dev = raw.cuda()
    ms_dev = timed(lambda: mapper(dev, K), reps=200)
    ms_host = timed(lambda: mapper(raw, K), reps=50)
    d = mapper(dev, K)
    print(f"f3 DeviceInputMapper 370x1224 -> {tuple(d['image'].shape[1:])}: {ms_dev * 1e3:7.1f} us per image with the raw frame resident "
          f"({1e3 / ms_dev:8.0f} img/s), {ms_host * 1e3:7.1f} us from pageable host memory (1.36 MB H2D included)")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    mapper_line()  # (first: the launch plans of the TTA lines below take 14+ GB; the small allocations of this loop then measure the allocator)
    eval_lines()
    tta_lines()
    nusc_tta_line()
