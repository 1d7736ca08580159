"""Context number for BASELINE.md (build container only): the REAL reference forward -- tridet.modeling.dd3d.core.DD3D from /root/reference over
the third-party shims of ref_shims.py -- timed on this container's CPU on the bench workload (one synthetic 384x1280 KITTI-shaped image),
beside the oracle (what bench.py's `cpu_baseline` times on the GPU box, where /root/reference does not exist).

    python tests/golden/time_reference.py [forwards]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from tests.golden.make_golden import TRAINING_ONLY_KEYS, build_reference_model, case_inputs  # noqa: E402


def main():
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.synthetic import load_calib, make_state_dict
    from oracle import dd3d_oracle as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    threads = torch.get_num_threads()
    cfg = get_cfg("dd3d_kitti_dla34", dict(TRAINING_ONLY_KEYS))
    ours = META_ARCH_REGISTRY.get("DD3D")(cfg)
    sd = make_state_dict(ours, calib=load_calib("dla34_kitti"))
    ref = build_reference_model(cfg)
    ref.load_state_dict(sd, strict=True)
    inputs = case_inputs(1, 384, 1280, False, "kitti", reference_pose=True)
    with torch.no_grad():
        ref(inputs)
        t0 = time.perf_counter()
        for _ in range(n):
            out = ref(inputs)
        t_ref = (time.perf_counter() - t0) / n
        O.dd3d_forward(sd, cfg, inputs)
        t0 = time.perf_counter()
        for _ in range(n):
            res, _ = O.dd3d_forward(sd, cfg, inputs)
        t_or = (time.perf_counter() - t0) / n
    print(f"reference tridet DD3D.forward (CPU, {threads} torch threads, shimmed third-party packages): {t_ref * 1e3:.1f} ms / image = {1 / t_ref:.2f} img/s "
          f"({len(out[0]['instances'])} detections);  oracle on the same inputs: {t_or * 1e3:.1f} ms = {1 / t_or:.2f} img/s ({len(res[0]['scores'])} detections)")


if __name__ == "__main__":
    main()
