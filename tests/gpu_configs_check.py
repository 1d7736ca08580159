"""Dev tool: run the other BASELINE.json configurations at full size through the HIP path (graph replay) and report time,
detections and peak memory:  DD3D-V2-99 KITTI 384x1280 (B = 1 and 4), NuscenesDD3D-DLA34 896x1600 (one 6-camera sample).

    python tests/gpu_configs_check.py [substring of the experiment name, e.g. v99]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402

CASES = [("dd3d_kitti_v99", "v99_kitti", "kitti", 1, 384, 1280), ("dd3d_kitti_v99", "v99_kitti", "kitti", 4, 384, 1280),
         # BASELINE.json configs[2]: DD3D-V2-99 KITTI3D 384x1280 bs=16
         ("dd3d_kitti_v99", "v99_kitti", "kitti", 16, 384, 1280),
         ("dd3d_nusc_dla34", "dla34_nusc", "nusc", 6, 896, 1600), ("dd3d_kitti_dla34", "dla34_kitti", "kitti", 8, 384, 1280),
         # BASELINE.json configs[3] per GPU: V2-99 on one 6-camera nuScenes sample (900 x 1600 -> 896 x 1593, padded to /64)
         ("dd3d_nusc_v99", "v99_nusc", "nusc", 6, 896, 1600)]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    only_b = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for exp, tag, ds, B, H, W in CASES:
        if only not in exp or (only_b and B != only_b):
            continue
        torch.cuda.reset_peak_memory_stats()
        cfg = get_cfg(exp)
        model = build_model(cfg)
        model.load_state_dict(make_state_dict(model, calib=load_calib(tag)))
        inputs = make_inputs(B, H, W, dataset=ds)
        t0 = time.time()
        out = model(inputs)
        torch.cuda.synchronize()
        t_build = time.time() - t0
        plan = next(iter(model._plans.values()))
        for _ in range(3):
            plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 10
        for _ in range(n):
            plan.run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / n
        gmac = plan.conv_macs / 1e9
        print(f"{exp:18s} math={os.environ.get('DD3D_MATH', 'f16x2'):7s} B={B} {H}x{W}: {ms:8.3f} ms/forward = {B / ms * 1e3:7.1f} img/s, {2 * gmac / ms:7.1f} TFLOP/s, dets {[len(o['instances']) for o in out]}, "
              f"plan build {t_build:.1f} s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, ops {len(plan.ops)}", flush=True)
        del model, plan, out
        import gc
        gc.collect()  # (plans and their ops reference each other: without a collection the previous case's buffers inflate the next peak)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
