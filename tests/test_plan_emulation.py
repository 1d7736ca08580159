"""Host side of the engine for configurations beyond the ones benchmarked: the launch plan (dry run, CPU buffers) is executed by the
torch emulator of tests/plan_emulator.py -- packed filters, folded norms, concat-by-placement, residual wiring, pooling / top-down /
eSE ops exactly as the plan describes them -- and every backbone stage, FPN level and head map must equal the oracle's (which is pinned
against the reference).  Covers VoVNet specs no GPU test has run yet."""
import pytest
import torch

from tests.plan_emulator import emulate

CASES = {
    "dla34_kitti": ("dd3d_kitti_dla34", "dla34_kitti", None, "kitti", 1, 128, 256),
    "dla34_nusc": ("dd3d_nusc_dla34", "dla34_nusc", None, "nusc", 6, 128, 128),  # one complete 6-camera sample
    "v99_kitti": ("dd3d_kitti_v99", "v99_kitti", None, "kitti", 1, 64, 128),
    "v39_kitti": ("dd3d_kitti_v99", "v99_kitti", {"FE": {"BACKBONE": {"NAME": "V-39-eSE"}}}, "kitti", 1, 64, 128),
    "v19_kitti": ("dd3d_kitti_v99", "v99_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-eSE"}}}, "kitti", 1, 64, 128),
    "v57_kitti": ("dd3d_kitti_v99", "v99_kitti", {"FE": {"BACKBONE": {"NAME": "V-57-eSE"}}}, "kitti", 1, 64, 128),
    # 64 / 80 / 96 / 112-channel layers: every slice of the OSA concat buffers padded to a 32-channel boundary, filters scattered accordingly
    "v19slim_kitti": ("dd3d_kitti_v99", "v19slim_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-slim-eSE"}}}, "kitti", 1, 64, 128),
    # depthwise-separable layers: the depthwise 3x3 runs as a dense convolution with a diagonal filter
    "v19dw_kitti": ("dd3d_kitti_v99", "v19dw_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-dw-eSE"}}}, "kitti", 1, 64, 128),
    "v19slimdw_kitti": ("dd3d_kitti_v99", "v19slimdw_kitti", {"FE": {"BACKBONE": {"NAME": "V-19-slim-dw-eSE"}}}, "kitti", 1, 64, 128),
    "dla34_plain_heads": ("dd3d_kitti_dla34", "dla34_kitti",
                          {"DD3D": {"FCOS2D": {"USE_SCALE": False}, "FCOS3D": {"USE_SCALE": False, "PER_LEVEL_PREDICTORS": True}}}, "kitti", 1, 128, 256),
    "dla34_box2d_only": ("dd3d_kitti_dla34", "dla34_kitti", {"MODEL": {"BOX3D_ON": False}}, "kitti", 1, 128, 256),
    "dla34_class_agnostic": ("dd3d_kitti_dla34", "dla34_kitti", {"DD3D": {"FCOS3D": {"CLASS_AGNOSTIC_BOX3D": True}}}, "kitti", 1, 128, 256),
    "v99_nusc": ("dd3d_nusc_v99", "v99_nusc", None, "nusc", 6, 64, 128),  # BASELINE.json configs[3]'s architecture
    "dla34_ragged": ("dd3d_kitti_dla34", "dla34_kitti", None, "ragged", 2, 128, 384),
    "dla34_fpn_without_norm": ("dd3d_kitti_dla34", "dla34_kitti", {"FE": {"FPN": {"NORM": ""}}}, "kitti", 1, 128, 256),
    "dla34_swapped_head_norms": ("dd3d_kitti_dla34", "dla34_kitti", {"DD3D": {"FCOS2D": {"NORM": "FrozenBN"}, "FCOS3D": {"NORM": "BN"}}},
                                 "kitti", 1, 128, 256),
    "dla34_bn_backbone": ("dd3d_kitti_dla34", "dla34_kitti", {"FE": {"BACKBONE": {"NORM": "BN"}}}, "kitti", 1, 128, 256),
    "dla34_odd_towers": ("dd3d_kitti_dla34", "dla34_kitti", {"FE": {"FPN": {"OUT_CHANNELS": 128}},
                                                             "DD3D": {"NUM_CLASSES": 3, "FCOS2D": {"NUM_CLS_CONVS": 2, "NUM_BOX_CONVS": 3},
                                                                      "FCOS3D": {"NUM_CONVS": 1}}}, "kitti", 1, 128, 256),
    "dla34_three_levels": ("dd3d_kitti_dla34", "dla34_kitti", {"DD3D": {"IN_FEATURES": ["p3", "p4", "p5"]}}, "kitti", 1, 128, 256),
    # Bottleneck DLA variants: deeper trees (up to five levels), residual roots -- the generic tree lowering
    "dla46c_kitti": ("dd3d_kitti_dla34", "dla46c_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-46-C"}}}, "kitti", 1, 128, 256),
    "dla60_kitti": ("dd3d_kitti_dla34", "dla60_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-60"}}}, "kitti", 1, 128, 256),
    "dla102_kitti": ("dd3d_kitti_dla34", "dla102_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-102"}}}, "kitti", 1, 128, 256),
    "dla169_kitti": ("dd3d_kitti_dla34", "dla169_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-169"}}}, "kitti", 1, 128, 256),
    # BottleneckX variants: the grouped 3x3 runs as a dense convolution with a block-diagonal filter
    "dlax46c_kitti": ("dd3d_kitti_dla34", "dlax46c_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-46-C"}}}, "kitti", 1, 128, 256),
    "dlax60c_kitti": ("dd3d_kitti_dla34", "dlax60c_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-60-C"}}}, "kitti", 1, 128, 256),
    "dlax60_kitti": ("dd3d_kitti_dla34", "dlax60_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-60"}}}, "kitti", 1, 128, 256),
    "dlax102_kitti": ("dd3d_kitti_dla34", "dlax102_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-102"}}}, "kitti", 1, 128, 256),
    "dlax10264_kitti": ("dd3d_kitti_dla34", "dlax10264_kitti", {"FE": {"BACKBONE": {"NAME": "DLA-X-102-64"}}}, "kitti", 1, 128, 256),
}


# Backbone specs SURVEY section 2 marks OUT OF SCOPE (only DLA-34 and V2-99 are on the hot path): their emulation cases cost four minutes of
# CPU and are opt-in (DD3D_TEST_VARIANTS=1; tests/conftest.py) -- the round-4 verdict asked for no more time there.
OUT_OF_SCOPE = {n for n in CASES if n.split("_")[0] in ("v39", "v19", "v57", "v19slim", "v19dw", "v19slimdw", "dla46c", "dla60", "dla102", "dla169",
                                                         "dlax46c", "dlax60c", "dlax60", "dlax102", "dlax10264")}


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.variants) if n in OUT_OF_SCOPE else n for n in CASES])
def test_emulated_plan_matches_oracle(hiplib, name):
    from dd3d_amd import META_ARCH_REGISTRY
    from dd3d_amd.engine import ForwardPlan
    from dd3d_amd.synthetic import make_inputs
    from oracle import dd3d_oracle as O
    from oracle import nuscenes_oracle as N
    from tests.util import bundle
    exp, tag, over, ds, B, H, W = CASES[name]
    cfg, sd = bundle(exp, tag, over)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.load_state_dict(sd, strict=True)
    div = model.backbone.size_divisibility
    if ds == "ragged":  # images of different sizes in one padded batch (image_list.py:120-142)
        inputs = make_inputs(1, H, W) + make_inputs(1, H - 37, W - 101, seed=7)
    else:
        inputs = make_inputs(B, H, W, dataset=ds)
    plan = ForwardPlan(model, B, H + (-H) % div, W + (-W) % div, device="cpu", dry_run=True)
    model.stage_inputs(inputs, plan=plan)
    with torch.no_grad():
        done = emulate(plan)
        if cfg.MODEL.META_ARCHITECTURE == "NuscenesDD3D":
            _, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
        else:
            _, st = O.dd3d_forward(sd, cfg, inputs, stop_after_heads=True)
    assert "predictors" in done or "predictors.narrow" in done  # (every group of <= 32 channels: one launch on the 32-column tile)

    def close(got, ref, what):
        err = float((got - ref).abs().max())
        assert got.shape == ref.shape and err < 1e-4 * max(1.0, float(ref.abs().max())), (what, err)

    close(plan.bufs["img4"].nchw(0, 3), st["images"], "images")
    for k, v in st.get("bottom_up", {}).items():
        if k in plan.bottom_up:
            got = plan.bottom_up[k].nchw()  # the buffer may carry zero channels up to the next multiple of 32
            assert float(got[:, v.shape[1]:].abs().max() if got.shape[1] > v.shape[1] else 0.0) == 0.0
            close(got[:, :v.shape[1]], v, k)
    C = cfg.DD3D.NUM_CLASSES
    for l in range(len(st["features"])):
        close(plan.features[l].nchw(), st["features"][l], f"feature {l}")
        close(plan.cls_maps[l].nchw(0, C), st["logits"][l], f"logits {l}")
        close(plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l], f"box2d_reg {l}")
        close(plan.b2d_maps[l].nchw(4, 1), st["centerness"][l], f"centerness {l}")
        if "quat" in st:
            fused = torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1)
            close(plan.b3d_maps[l].nchw(0, fused.shape[1]), fused, f"box3d {l}")
        if "attr" in st:
            na = st["attr"][l].shape[1]
            close(plan.cls_maps[l].nchw(C, na), st["attr"][l], f"attr {l}")
            close(plan.cls_maps[l].nchw(C + na, 1), st["speed"][l], f"speed {l}")


def test_unbuilt_vovnet_spec_fails_loudly():
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    import dd3d_amd.modeling  # noqa: F401
    cfg = get_cfg("dd3d_kitti_v99", {"FE": {"BACKBONE": {"NAME": "V-27-eSE"}}})
    with pytest.raises(NotImplementedError, match="unknown VoVNet spec"):
        META_ARCH_REGISTRY.get("DD3D")(cfg)
    cfg = get_cfg("dd3d_kitti_dla34", {"FE": {"BACKBONE": {"NAME": "DLA-35"}}})
    with pytest.raises(NotImplementedError, match="unknown DLA variant"):
        META_ARCH_REGISTRY.get("DD3D")(cfg)


@pytest.mark.parametrize("spec", ["V-19-eSE", "V-39-eSE", "V-57-eSE", "fpn-without-norm", "swapped-head-norms", "bn-backbone", "odd-towers", "three-levels", "DLA-46-C", "DLA-60", "DLA-102", "DLA-169", "DLA-X-46-C", "DLA-X-60-C", "DLA-X-60", "DLA-X-102", "DLA-X-102-64", "V-19-slim-eSE", "V-19-dw-eSE", "V-19-slim-dw-eSE"])
def test_oracle_vovnet_specs_match_reference_golden(spec):
    """The oracle the emulated plans are compared with is itself pinned for these construction variants: compact goldens from the
    reference's own backbone + FPN + heads (tests/golden/make_golden.py vovnet_specs)."""
    import os
    import numpy as np
    from oracle import dd3d_oracle as O
    from tests.golden.make_golden import STRUCTURAL, case_inputs
    from tests.util import bundle
    exp, tag, over = STRUCTURAL[spec]
    cfg, sd = bundle(exp, tag, over)
    H, W = (64, 128) if "v99" in exp else (128, 256)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"vovnet_spec_{spec.replace('-', '').lower()}.npz"))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    with torch.no_grad():
        res, st = O.dd3d_forward(sd, cfg, case_inputs(1, H, W, False, "kitti"))
    levels = [l for l in range(len(st["features"])) if f"feat{l}" in g]
    assert len(levels) >= 2
    for l in levels:
        assert torch.allclose(st["features"][l], t(f"feat{l}"), rtol=1e-5, atol=1e-5)
        assert torch.allclose(st["logits"][l], t(f"logits{l}"), rtol=1e-5, atol=2e-5) and torch.allclose(st["depth"][l], t(f"depth{l}"), rtol=1e-5, atol=2e-5)
    r = res[0]
    assert len(r["scores"]) == len(g["det0_scores_3d"])
    if len(r["scores"]):
        assert torch.equal(r["pred_classes"], t("det0_classes")) and torch.equal(r["locations"], t("det0_locations"))
        assert torch.allclose(r["pred_boxes"], t("det0_boxes"), rtol=1e-5, atol=1e-4) and torch.allclose(r["scores_3d"], t("det0_scores_3d"), rtol=1e-5)
        assert torch.allclose(r["pred_boxes3d"]["depth"], t("det0_depth"), rtol=1e-5)


@pytest.mark.parametrize("offset,by_focal,use_scale,convs", [("none", True, True, 4), ("half", False, True, 1), ("half", True, False, 4), ("none", False, False, 1)])
def test_emulated_dense_depth_plan_matches_oracle(hiplib, offset, by_focal, use_scale, convs):
    """(The oracle agrees with the reference's DD3DDenseDepth on all 16 combinations of these switches: fuzz_reference.py dense_depth.)"""
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.engine import DenseDepthPlan
    from dd3d_amd.synthetic import load_calib, make_state_dict
    from oracle import dense_depth_oracle as D
    from tests.test_dense_depth import OVER, _case
    _, _, inputs = _case()
    cfg = get_cfg("dd3d_kitti_dla34", {"MODEL": OVER["MODEL"], "DD3D": dict(OVER["DD3D"], FEATURE_LOCATIONS_OFFSET=offset, FCOS3D={
        "SCALE_DEPTH_BY_FOCAL_LENGTHS": by_focal, "USE_SCALE": use_scale, "NUM_CONVS": convs})})
    sd = make_state_dict(META_ARCH_REGISTRY.get("DD3DDenseDepth")(cfg), calib=load_calib("dla34_kitti"))
    model = META_ARCH_REGISTRY.get("DD3DDenseDepth")(cfg)
    model.load_state_dict(sd, strict=True)
    plan = DenseDepthPlan(model, 2, 128, 256, device="cpu", dry_run=True)
    for i, x in enumerate(inputs):  # the staging part of DD3DDenseDepth.predict_dense_depth
        plan.in_u8[i, :, :x["image"].shape[1], :x["image"].shape[2]].copy_(x["image"])
    plan.in_sizes.copy_(torch.tensor([[int(x["image"].shape[-2]), int(x["image"].shape[-1])] for x in inputs], dtype=torch.int32))
    plan.in_K.copy_(torch.stack([x["intrinsics"].float() for x in inputs], 0).reshape(2, 9))
    with torch.no_grad():
        done = emulate(plan, stop_before=())
        maps, _ = D.dense_depth_forward(sd, cfg, inputs)
    assert done[-1] == "dd_upsample.4" and len(maps) == len(plan.depth_maps) == 5
    for l, (got, ref) in enumerate(zip(plan.depth_maps, maps)):
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), l


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 1000, 1001, 1002, 1003, 1004, 1005, 1006, 1007])
def test_emulated_plan_matches_oracle_on_random_switch_combinations(hiplib, seed):
    """Cross product of the construction switches (same generator as tests/golden/fuzz_reference.py, which checks the oracle against the
    reference itself over these combinations): the dry-run plan executed on the CPU must reproduce the oracle's head maps."""
    import random
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.engine import ForwardPlan
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    from oracle import dd3d_oracle as O
    from oracle import nuscenes_oracle as N
    from tests.golden.fuzz_reference import random_case
    import dd3d_amd.modeling  # noqa: F401
    exp, tag, over, nusc, v99 = random_case(random.Random(100 + seed), backbones=seed >= 1000)  # seeds >= 1000 also draw the backbone variant
    cfg = get_cfg(exp, over)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    sd = make_state_dict(model, calib=load_calib(tag))
    model.load_state_dict(sd, strict=True)
    B, H, W = (6, 64, 128) if nusc else (1, 64 if v99 else 128, 128 if v99 else 256)
    inputs = make_inputs(B, H, W, dataset="nusc" if nusc else "kitti")
    div = model.backbone.size_divisibility
    plan = ForwardPlan(model, B, H + (-H) % div, W + (-W) % div, device="cpu", dry_run=True)
    model.stage_inputs(inputs, plan=plan)
    with torch.no_grad():
        emulate(plan)
        _, st = N.nuscenes_dd3d_forward(sd, cfg, inputs) if nusc else O.dd3d_forward(sd, cfg, inputs, stop_after_heads=True)
    C = cfg.DD3D.NUM_CLASSES

    def close(got, ref, what):
        err = float((got - ref).abs().max())
        assert got.shape == ref.shape and err < 1e-4 * max(1.0, float(ref.abs().max())), (what, err, over)

    for l in range(len(st["features"])):
        close(plan.features[l].nchw(), st["features"][l], f"feature {l}")
        close(plan.cls_maps[l].nchw(0, C), st["logits"][l], f"logits {l}")
        close(plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l], f"box2d_reg {l}")
        close(plan.b2d_maps[l].nchw(4, 1), st["centerness"][l], f"centerness {l}")
        if "quat" in st:
            fused = torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1)
            close(plan.b3d_maps[l].nchw(0, fused.shape[1]), fused, f"box3d {l}")
    a = plan.select_args
    inf, c3 = cfg.DD3D.FCOS2D.INFERENCE, cfg.DD3D.FCOS3D
    assert (a.thresh_with_ctr, a.topk, a.loc_offset_half) == (int(inf.THRESH_WITH_CTR), int(inf.PRE_NMS_TOPK), int(cfg.DD3D.FEATURE_LOCATIONS_OFFSET == "half"))
    assert abs(a.pre_nms_thresh - float(inf.PRE_NMS_THRESH)) < 1e-7
    if cfg.MODEL.BOX3D_ON:
        assert (a.class_agnostic_3d, a.allocentric, a.depth_is_distance, a.scale_depth_by_focal) == (
            int(c3.CLASS_AGNOSTIC_BOX3D), int(c3.PREDICT_ALLOCENTRIC_ROT), int(c3.PREDICT_DISTANCE), int(c3.SCALE_DEPTH_BY_FOCAL_LENGTHS))


def test_generic_tree_lowering_reproduces_dla34(hiplib):
    """The depth-generic DLA tree lowering (used for the Bottleneck variants) applied to DLA-34 gives the same features as the oracle, like
    the side-branched lowering DLA-34 normally takes."""
    from dd3d_amd import META_ARCH_REGISTRY
    from dd3d_amd.engine import ForwardPlan
    from dd3d_amd.synthetic import make_inputs
    from oracle import dd3d_oracle as O
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti", None)
    model = META_ARCH_REGISTRY.get("DD3D")(cfg)
    model.load_state_dict(sd, strict=True)
    model.force_generic_dla = True
    inputs = make_inputs(1, 128, 256)
    plan = ForwardPlan(model, 1, 128, 256, device="cpu", dry_run=True)
    assert any(op.name.endswith("level3.tree2.root") for op in plan.ops) and not plan._side_streams
    model.stage_inputs(inputs, plan=plan)
    with torch.no_grad():
        emulate(plan)
        _, st = O.dd3d_forward(sd, cfg, inputs, stop_after_heads=True)
    for l in range(5):
        ref = st["features"][l]
        assert float((plan.features[l].nchw() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), l


@pytest.mark.parametrize("math", ["f32", "bf16x3", "f16x2", "bf16x2", "bf16", "bf16x3-f32in"])
@pytest.mark.parametrize("name", ["dla34_kitti", "v99_kitti", "dla34_nusc"])
def test_storage_forms_are_consistent_in_every_math_mode(hiplib, name, math, monkeypatch):
    """Every arithmetic mode lays its activations out differently (f32 NHWC only; split planes of 3 / 2 / 1 terms next to or instead
    of f32).  The emulator tracks which storage of which channels each op writes and checks every read against it: a convolution
    streaming the planes of a tensor only its f32 side was written to would read stale memory on the device."""
    from dd3d_amd import META_ARCH_REGISTRY
    from dd3d_amd.engine import ConvOp, ForwardPlan
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle
    exp, tag, over, ds, B, H, W = CASES[name]
    cfg, sd = bundle(exp, tag, over)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.load_state_dict(sd, strict=True)
    if math == "bf16x3-f32in":
        monkeypatch.setenv("DD3D_PLANES", "0")
        math = "bf16x3"
    model.math = math
    div = model.backbone.size_divisibility
    plan = ForwardPlan(model, B, H + (-H) % div, W + (-W) % div, device="cpu", dry_run=True)
    model.stage_inputs(make_inputs(B, H, W, dataset=ds), plan=plan)
    with torch.no_grad():
        done = emulate(plan)
    assert "predictors" in done or "predictors.narrow" in done  # (every group of <= 32 channels: one launch on the 32-column tile)
    forms = {op.info["in_form"] for op in plan.ops if isinstance(op, ConvOp) and op.L.Cin % 32 == 0}
    assert forms == ({"planes"} if plan.use_planes else {"f32"}), forms
