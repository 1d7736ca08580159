"""Deterministic synthetic weights and inputs (there are no datasets / checkpoints offline).

SURVEY.md section 8d: seeded reference-style initialisation, *randomised* (Frozen)BN statistics so that norm folding
is really exercised, and classifier gains calibrated once (offline, tests/golden/calibrate_synthetic.py ->
dd3d_amd/data/synth_calib_*.json) so that roughly 1 % of the (location, class) scores pass PRE_NMS_THRESH --
otherwise a random network yields no candidates (or all of them) and decode / NMS are not exercised.

Used by bench.py, __graft_entry__.smoke() and the tests; the same state_dict feeds the HIP path and the oracle.
"""
import json
import os

import torch

_DATA_DIR = os.path.join(os.path.dirname(__file__), "data")

KITTI_K = [[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]]  # KITTI cam-2 intrinsics
NUSC_K = [[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]]


def calib_path(tag):
    """The package ships the calibrations of the four benchmarked / tested configurations (DLA-34 and V2-99, KITTI and nuScenes); the
    other backbone specs' files live with the tests that emulate them (tests/data, named by DD3D_CALIB_DIR -- tests/conftest.py)."""
    p = os.path.join(_DATA_DIR, f"synth_calib_{tag}.json")
    extra = os.environ.get("DD3D_CALIB_DIR")
    if not os.path.exists(p) and extra and os.path.exists(os.path.join(extra, f"synth_calib_{tag}.json")):
        return os.path.join(extra, f"synth_calib_{tag}.json")
    return p


SHIPPED_CALIBS = ("dla34_kitti", "dla34_nusc", "v99_kitti", "v99_nusc")


def load_calib(tag, required=False):
    """Calibration of one synthetic configuration ({} when there is none: the state dict is then uncalibrated and a random network
    yields no candidates -- or all of them).  `required`: raise instead, naming where the file is expected."""
    p = calib_path(tag)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    if required:
        raise FileNotFoundError(
            f"no synthetic calibration '{tag}': the package ships {', '.join(SHIPPED_CALIBS)} (dd3d_amd/data/synth_calib_*.json); the other "
            "backbone specs' files live under tests/data and are found through DD3D_CALIB_DIR (tests/conftest.py sets it) -- "
            f"expected {os.path.basename(p)} in {_DATA_DIR} or $DD3D_CALIB_DIR={os.environ.get('DD3D_CALIB_DIR')!r}")
    return {}


def make_state_dict(model, seed=0, calib=None):
    """Return a CPU state_dict for ``model`` (any dd3d_amd meta-arch): conv weights keep the model's own seeded
    initialisation (re-drawn here under ``seed``), every norm gets random affine + statistics.
    ``calib`` maps a norm prefix -> [mean, std] of its input activation (measured once with the oracle) and the
    predictor prefixes -> [gain, bias]."""
    from dd3d_amd.layers import BatchNorm2d, Conv2d, FrozenBatchNorm2d
    calib = calib or {}
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for name, mod in model.named_modules():
        if isinstance(mod, Conv2d):
            w = sd[name + ".weight"]
            fan_out = w.shape[0] * w.shape[2] * w.shape[3]
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            is_pred = mod.norm is None and mod.bias is not None and name.split(".")[0] in ("fcos2d_head", "fcos3d_head") \
                or name in ("attr_logits", "speed") or ".box3d_depth." in name
            if is_pred:  # kaiming_uniform_(a=1): U(-sqrt(3/fan_in), +)
                bound = (3.0 / fan_in)**0.5
                w.copy_((torch.rand(w.shape, generator=g) * 2 - 1) * bound)
            else:  # kaiming_normal_(fan_out, relu)
                w.copy_(torch.randn(w.shape, generator=g) * (2.0 / fan_out)**0.5)
            if name + ".bias" in sd:
                sd[name + ".bias"].copy_(torch.randn(w.shape[0], generator=g) * 0.02)
            if name in calib:  # predictor gain / bias
                gain, bias = calib[name]
                w.mul_(gain)
                if name + ".bias" in sd:
                    sd[name + ".bias"].add_(bias)
        elif isinstance(mod, (BatchNorm2d, FrozenBatchNorm2d)):
            n = mod.num_features
            m0, s0, gain = (list(calib.get(name, [0.0, 1.0])) + [1.0])[:3]  # gain: per-level equalisation of the logits
            sd[name + ".weight"].copy_((torch.rand(n, generator=g) + 0.5) * gain)
            sd[name + ".bias"].copy_(torch.randn(n, generator=g) * 0.1 * gain)
            sd[name + ".running_mean"].copy_(m0 + torch.randn(n, generator=g) * 0.1 * s0)
            sd[name + ".running_var"].copy_((torch.rand(n, generator=g) + 0.5) * s0 * s0)
    return sd


def make_inputs(B=1, H=384, W=1280, dataset="kitti", seed=1000, out_hw=None, device="cpu"):
    """``batched_inputs`` with the schema of DefaultDatasetMapper (tridet/data/dataset_mappers/dataset_mapper.py:100-201):
    uint8 (3,H,W) BGR ``image``, 3x3 float32 ``intrinsics`` (already adjusted for the resize), ``height``/``width``."""
    if dataset == "kitti":
        K = torch.tensor(KITTI_K) * torch.tensor([[1270.0 / 1224.0], [384.0 / 370.0], [1.0]])
    else:
        K = torch.tensor(NUSC_K) * torch.tensor([[1593.0 / 1600.0], [896.0 / 900.0], [1.0]])
    out = []
    for i in range(B):
        g = torch.Generator().manual_seed(seed + i)
        img = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
        d = {"image": img.to(device), "intrinsics": K.clone().float(), "height": H, "width": W, "image_id": i,
             "file_name": f"synthetic_{seed + i}.png"}
        if out_hw is not None:
            d["height"], d["width"] = out_hw
        if dataset != "kitti":
            # nuScenes-shaped extras (dataset_mapper.py:155-165): 6 cameras per sample (yaw 0, +-55, +-110, 180 deg around the
            # vertical axis of the ego frame, composed with the camera-to-vehicle axis swap) times one ego pose per sample
            from dd3d_amd.structures import Pose
            cam, sample = i % 6, i // 6
            yaw = [0.0, 55.0, -55.0, 110.0, -110.0, 180.0][cam]
            cam_to_vehicle = Pose((0.5, -0.5, 0.5, -0.5), (1.5, 0.2 * (cam - 2.5), 1.6))  # z fwd, x right, y down -> x fwd, y left, z up
            ego = Pose.from_yaw(17.0 * sample + 5.0, (410.0 + 13.0 * sample, 1180.0 - 7.0 * sample, 0.0))
            d["pose"] = ego * Pose.from_yaw(yaw) * cam_to_vehicle
            d["sample_token"] = f"s{sample}"
        out.append(d)
    return out
