"""Parity of the HIP forward path (through the C ABI) against the CPU oracle.

Tolerances (BASELINE.json north_star): class / index integers bit-exact, box / depth floats <= 1e-3 relative.
Selection steps (score > 0.05, top-k, IoU > thr) are discontinuous, so integer exactness is asserted on IDENTICAL head
maps (the oracle's maps are fed to the HIP post-processing); the end-to-end runs assert float parity of the head maps
and of every detection that both sides produce, and report how many candidates sit within 1e-4 of a threshold.
"""
import pytest
import torch

from tests.util import candidate_margins, candidates_from_plan, gpu_model, max_abs, oracle_heads_to_plan, quat_err, rel_err

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3  # north_star: box/depth floats within 1e-3 rel
MARGIN_EPS = 2e-6  # |oracle score - cut| of a candidate only one side selected (scores are in (0.05, 1): ~1e-5 relative)


def _oracle(cfg, sd, inputs):
    from oracle import dd3d_oracle as O
    with torch.no_grad():
        return O.dd3d_forward(sd, cfg, inputs)


def _key(levels, locs, classes):
    return [(int(a), float(x), float(y), int(c)) for a, (x, y), c in zip(levels.tolist(), locs.tolist(), classes.tolist())]


def _check_head_maps(plan, st, C):
    for l in range(len(st["logits"])):
        assert rel_err(plan.features[l].nchw(), st["features"][l]) < 2e-2  # |x| << max entries: abs error is what matters
        assert max_abs(plan.features[l].nchw(), st["features"][l]) < 1e-4 * float(st["features"][l].abs().max())
        for name, got, ref in [
            ("logits", plan.cls_maps[l].nchw(0, C), st["logits"][l]), ("box2d_reg", plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l]),
            ("centerness", plan.b2d_maps[l].nchw(4, 1), st["centerness"][l]),
            ("box3d", plan.b3d_maps[l].t[..., :11 * C].permute(0, 3, 1, 2),
             torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1))
        ]:
            e = max_abs(got, ref)
            assert e < 1e-4 * max(1.0, float(ref.abs().max())), (name, l, e)


def _check_final(out, ref, exact_ints=True):
    o = out["instances"]
    assert len(o) == len(ref["scores"])
    if len(o) == 0:
        return
    if exact_ints:
        assert torch.equal(o.pred_classes.cpu(), ref["pred_classes"])
        assert torch.equal(o.fpn_levels.cpu(), ref["fpn_levels"])
        assert torch.equal(o.locations.cpu(), ref["locations"])
    assert max_abs(o.pred_boxes.tensor, ref["pred_boxes"]) <= REL_TOL * max(1.0, float(ref["pred_boxes"].abs().max()))
    assert rel_err(o.scores, ref["scores"]) < REL_TOL and rel_err(o.scores_3d, ref["scores_3d"]) < REL_TOL
    b = ref["pred_boxes3d"]
    assert rel_err(o.pred_boxes3d.depth, b["depth"]) < REL_TOL
    assert rel_err(o.pred_boxes3d.size, b["size"]) < REL_TOL
    assert max_abs(o.pred_boxes3d.proj_ctr, b["proj_ctr"]) <= REL_TOL * max(1.0, float(b["proj_ctr"].abs().max()))
    assert quat_err(o.pred_boxes3d.quat, b["quat"]) < REL_TOL
    from oracle import dd3d_oracle as O
    tv = O.boxes3d_tvec(b)
    assert max_abs(o.pred_boxes3d.tvec, tv) <= REL_TOL * max(1.0, float(tv.abs().max()))
    # 3D-box L1 on the 8 corners (sign-free check of the rotation)
    c_ref = O.boxes3d_corners(b["quat"], tv, b["size"])
    c_got = o.pred_boxes3d.to("cpu").corners
    assert float((c_got - c_ref).abs().mean()) <= REL_TOL * max(1.0, float(c_ref.abs().mean()))


@pytest.mark.parametrize("H,W,B,math", [(128, 256, 2, None), (128, 256, 2, "bf16x3"), (128, 256, 2, "f32"), (384, 1280, 1, None), (384, 1280, 1, "bf16x3"),
                                        (384, 1280, 1, "f32")],
                         ids=["small_b2", "small_b2_bf16x3", "small_b2_f32mfma", "kitti_full", "kitti_full_bf16x3", "kitti_full_f32mfma"])
def test_forward_matches_oracle(hiplib, kitti_dla34, H, W, B, math):
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    model = gpu_model(cfg, sd, use_graph=False, math=math)
    inputs = make_inputs(B, H, W)
    if B > 1:  # ragged batch: second image smaller -> right/bottom zero padding after normalisation (image_list.py:120-142)
        inputs[1]["image"] = inputs[1]["image"][:, :H - 13, :W - 22].contiguous()
        inputs[1]["height"], inputs[1]["width"] = 99, 201  # rescaled output size
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    assert torch.equal(plan.normalized_image().cpu(), st["images"])  # normalise + pad is bit-exact
    _check_head_maps(plan, st, C)
    out = model.collect(plan, inputs, image_sizes)
    # end-to-end: a candidate only one side selected must sit ON a cut (oracle score within MARGIN_EPS of PRE_NMS_THRESH / of the
    # level's k-th score); with no such candidate the final detections are the same set; shared detections agree to REL_TOL
    for i in range(B):
        n_hip, n_ref, margins = candidate_margins(plan, st, cfg, i)
        print(f"[margin] image {i}: candidates hip={n_hip} oracle={n_ref} on-the-cut flips={len(margins)} "
              f"max distance from the cut={max(margins, default=0.0):.2e}")
        assert all(m <= MARGIN_EPS for m in margins), (n_hip, n_ref, margins)
        o, r = out[i]["instances"], ref[i]
        ko = _key(o.fpn_levels.cpu(), o.locations.cpu(), o.pred_classes.cpu())
        kr = _key(r["fpn_levels"], r["locations"], r["pred_classes"])
        common = set(ko) & set(kr)
        if not margins:
            assert ko == kr, (len(ko), len(kr), len(common))
        else:  # a flipped candidate may suppress / release its NMS neighbours: bounded by the flips, not by a blanket percentage
            assert len(set(ko) ^ set(kr)) <= 4 * len(margins), (len(ko), len(kr), len(common), len(margins))
        io = [ko.index(k) for k in common]
        ir = [kr.index(k) for k in common]
        if common:
            assert max_abs(o.pred_boxes.tensor[io], r["pred_boxes"][ir]) < REL_TOL * max(1.0, float(r["pred_boxes"].abs().max()))
            assert rel_err(o.pred_boxes3d.depth[io], r["pred_boxes3d"]["depth"][ir]) < REL_TOL
            assert rel_err(o.scores_3d[io], r["scores_3d"][ir]) < REL_TOL
    # integer-exact part: identical head maps in, identical candidates / detections out
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    for i in range(B):
        c, rc = candidates_from_plan(plan, i), st["candidates"][i]
        assert torch.equal(c["pred_classes"], rc["pred_classes"]) and torch.equal(c["fpn_levels"], rc["fpn_levels"])
        assert torch.equal(c["locations"], rc["locations"])
        assert c["counts"] == [len(info[i]["fg_inds"]) for info in st["level_info"]]
    out2 = model.collect(plan, inputs, image_sizes)
    for i in range(B):
        _check_final(out2[i], ref[i])


def test_topk_and_per_class_nms_paths(hiplib, kitti_dla34):
    """Low threshold => >1000 candidates on the big levels: exercises the radix-select top-k (fcos2d.py:309-317) and,
    with >1000 boxes per image, torchvision's per-class NMS branch (boxes.numel() > 4000)."""
    from dd3d_amd import get_cfg
    from dd3d_amd.synthetic import make_inputs
    _, _, sd = kitti_dla34
    cfg = get_cfg("dd3d_kitti_dla34", {"DD3D": {"FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.002, "PRE_NMS_TOPK": 300}}}})
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(1, 256, 512)
    ref, st = _oracle(cfg, sd, inputs)
    npass = [len(info[0]["fg_inds"]) for info in st["level_info"]]
    assert max(npass) > 300, npass  # the top-k branch is really taken
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    oracle_heads_to_plan(plan, st, cfg.DD3D.NUM_CLASSES)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    assert plan.npass[0].cpu().tolist() == npass
    c, rc = candidates_from_plan(plan, 0), st["candidates"][0]
    assert c["counts"] == [min(n, 300) for n in npass]
    assert len(rc["scores"]) > 1000  # per-class NMS branch
    # torch.topk(sorted=False) returns an arbitrary order: compare per-level candidate SETS
    ko = sorted(zip(c["fpn_levels"].tolist(), c["flat_index"].tolist()))
    flat_ref = []
    for l, info in enumerate(st["level_info"]):
        fg, cl, tk = info[0]["fg_inds"], info[0]["class_inds"], info[0]["topk_indices"]
        e = fg * cfg.DD3D.NUM_CLASSES + cl
        e = e[tk] if tk is not None else e
        flat_ref += [(l, int(v)) for v in e.tolist()]
    assert ko == sorted(flat_ref)
    out = model.collect(plan, inputs, image_sizes)[0]
    _check_final(out, ref[0])


def test_graph_replay_equals_eager_and_is_idempotent(hiplib, kitti_dla34):
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    inputs = make_inputs(1, 384, 1280)
    eager = gpu_model(cfg, sd, use_graph=False)
    graph = gpu_model(cfg, sd, use_graph=True)
    a = eager(inputs)[0]["instances"]
    b1 = graph(inputs)[0]["instances"]
    b2 = graph(inputs)[0]["instances"]
    for x, y in ((a, b1), (b1, b2)):
        assert len(x) == len(y) > 0
        assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor) and torch.equal(x.scores_3d, y.scores_3d)
        assert torch.equal(x.pred_classes, y.pred_classes) and torch.equal(x.pred_boxes3d.quat, y.pred_boxes3d.quat)
    # sortedness property of the NMS output (torchvision keep order = descending ranking score)
    assert bool((b1.scores_3d[:-1] >= b1.scores_3d[1:]).all())


def test_error_behaviour(hiplib, kitti_dla34):
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(1, 128, 256)
    inputs[0]["intrinsics"] = torch.eye(3)
    with pytest.raises(ValueError, match="Intrinsics is Identity"):
        model(inputs)


@pytest.mark.parametrize(
    "name", ["dla34_kitti_128x256_b1", "dla34_kitti_128x384_b2_ragged", "v99_kitti_128x256_b1", "dla34_nusc_128x224_b6", "dla34_nusc_128x224_b6_bevnms",
             "v99_nusc_64x128_b6"]
)
def test_hip_matches_reference_golden(hiplib, name):
    """HIP path vs the committed golden vectors (produced by the reference's own DD3D.forward, tests/golden/make_golden.py):
    head maps within float tolerance end-to-end; with the golden head maps as input, the HIP post-processing reproduces the
    reference's detections (classes / levels / locations bit-exact, floats within 1e-3 rel)."""
    import os
    import numpy as np
    from tests.golden.make_golden import CASES, DETECTIONS_ONLY, EXTRA_OVERRIDES, case_inputs
    from tests.util import bundle
    exp, tag, B, H, W, ragged = CASES[name]
    nusc = "nusc" in exp
    cfg, sd = bundle(exp, tag, EXTRA_OVERRIDES.get(name))
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, name + ".npz"))
    gh = np.load(os.path.join(gold, "dla34_nusc_128x224_b6.npz")) if name in DETECTIONS_ONLY else g  # same weights and inputs
    t = lambda k: torch.from_numpy(g[k] if k in g else gh[k])
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = case_inputs(B, H, W, ragged, "nusc" if nusc else "kitti")
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    assert torch.equal(plan.normalized_image().cpu(), t("images"))
    keys = ("logits", "box2d_reg", "centerness", "quat", "ctr", "depth", "size", "conf") + (("attr", "speed") if nusc else ())
    st = {k: [t(f"{k}{l}") for l in range(5)] for k in keys}
    for l in range(5):
        if nusc:
            na = st["attr"][l].shape[1]
            assert max_abs(plan.cls_maps[l].nchw(C, na), st["attr"][l]) < 1e-4 * max(1.0, float(st["attr"][l].abs().max()))
            assert max_abs(plan.cls_maps[l].nchw(C + na, 1), st["speed"][l]) < 1e-4 * max(1.0, float(st["speed"][l].abs().max()))
        # every head map of the HIP forward against the reference's own (golden) map, end to end
        C3 = st["quat"][l].shape[1] // 4  # C, or 1 for a class-agnostic 3D head
        for key, got in [("logits", plan.cls_maps[l].nchw(0, C)), ("box2d_reg", plan.b2d_maps[l].nchw(0, 4)), ("centerness", plan.b2d_maps[l].nchw(4, 1)),
                         ("quat", plan.b3d_maps[l].nchw(0, 4 * C3)), ("ctr", plan.b3d_maps[l].nchw(4 * C3, 2 * C3)), ("depth", plan.b3d_maps[l].nchw(6 * C3, C3)),
                         ("size", plan.b3d_maps[l].nchw(7 * C3, 3 * C3)), ("conf", plan.b3d_maps[l].nchw(10 * C3, C3))]:
            assert max_abs(got, st[key][l]) < 1e-4 * max(1.0, float(st[key][l].abs().max())), (name, key, l)
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    for i in range(B):
        o = out[i]["instances"]
        assert tuple(o.image_size) == tuple(g[f"det{i}_image_size"].tolist())
        assert torch.equal(o.pred_classes.cpu(), t(f"det{i}_classes")) and torch.equal(o.fpn_levels.cpu(), t(f"det{i}_levels"))
        assert torch.equal(o.locations.cpu(), t(f"det{i}_locations"))
        assert max_abs(o.pred_boxes.tensor, t(f"det{i}_boxes")) < REL_TOL * max(1.0, float(t(f"det{i}_boxes").abs().max()))
        assert rel_err(o.scores_3d, t(f"det{i}_scores_3d")) < REL_TOL and rel_err(o.pred_boxes3d.depth, t(f"det{i}_depth")) < REL_TOL
        assert rel_err(o.pred_boxes3d.size, t(f"det{i}_size")) < REL_TOL and quat_err(o.pred_boxes3d.quat, t(f"det{i}_quat")) < REL_TOL
        assert max_abs(o.pred_boxes3d.vectorize()[:, 4:], t(f"det{i}_vectorize")[:, 4:]) < REL_TOL * max(1.0, float(t(f"det{i}_vectorize").abs().max()))
        if nusc:
            assert torch.equal(o.pred_attributes.cpu(), t(f"det{i}_attributes")) and rel_err(o.pred_speeds, t(f"det{i}_speeds")) < REL_TOL
            gl, gg = o.pred_boxes3d_global.vectorize().cpu(), t(f"det{i}_global")
            assert max_abs(gl[:, 4:], gg[:, 4:]) < REL_TOL * max(1.0, float(gg[:, 4:].abs().max())) and quat_err(gl[:, :4], gg[:, :4]) < REL_TOL


def test_v99_forward_matches_oracle(hiplib):
    """DD3D-V2-99 (VoVNet-99-eSE + FPN P2..P6, BASELINE.json configs[2] architecture) at fp32: OSA concat-by-placement,
    ceil-mode 3x3 max-pool, eSE gate and identity add against the oracle; integer parity on identical head maps."""
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle
    cfg, sd = bundle("dd3d_kitti_v99", "v99_kitti")
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(2, 192, 320)
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    for k, v in st["bottom_up"].items():
        assert max_abs(plan.bottom_up[k].nchw(), v) < 1e-4 * max(1.0, float(v.abs().max())), k
    _check_head_maps(plan, st, C)
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    for i in range(2):
        _check_final(out[i], ref[i])


def _nusc_check(o, r, with_global):
    assert torch.equal(o.pred_attributes.cpu(), r["pred_attributes"]) and rel_err(o.pred_speeds, r["pred_speeds"]) < REL_TOL
    if with_global and len(o):
        gl, gg = o.pred_boxes3d_global.vectorize().cpu(), r["pred_boxes3d_global"]
        assert max_abs(gl[:, 4:], gg[:, 4:]) < REL_TOL * max(1.0, float(gg[:, 4:].abs().max())) and quat_err(gl[:, :4], gg[:, :4]) < REL_TOL


@pytest.mark.parametrize("bev_nms,cap", [(False, 500), (True, 45)], ids=["aggregate", "bevnms_cap45"])
def test_nuscenes_two_samples_match_oracle(hiplib, bev_nms, cap):
    """NuscenesDD3D on 2 samples x 6 cameras: attribute argmax / speed per candidate, resize, then the cross-camera BEV
    rotated NMS per sample (category = class + sample * C) with the batch-global cap, against the oracle on identical head
    maps (integer fields exact).  The 'bevnms' variant adds the per-image BEV NMS before the resize (DO_BEV_NMS)."""
    from oracle import nuscenes_oracle as N
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle
    over = {"DD3D": {"FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.01}}, "INFERENCE": {"DO_BEV_NMS": bev_nms},
                     "NUSC": {"INFERENCE": {"MAX_NUM_DETS_PER_SAMPLE": cap}}}}
    cfg, sd = bundle("dd3d_nusc_dla34", "dla34_nusc", over)
    model = gpu_model(cfg, sd, use_graph=False)
    B = 12
    inputs = make_inputs(B, 128, 160, dataset="nusc")
    for i, x in enumerate(inputs):
        x["height"], x["width"] = 128 + 8 * (i % 3), 300
    with torch.no_grad():
        ref, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    _check_head_maps(plan, st, C)
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    n_before = sum(len(x["scores"]) for x in st["before_aggregate"])
    n_after = sum(len(r["scores"]) for r in ref)
    assert n_after < n_before and (cap == 500 or n_after == cap)  # suppression (and the cap) really happen
    for i in range(B):
        _check_final(out[i], ref[i])
        _nusc_check(out[i]["instances"], ref[i], True)
    # graph replay of the same plan is idempotent
    model2 = gpu_model(cfg, sd, use_graph=True)
    a, b = model2(inputs), model2(inputs)
    for x, y in zip(a, b):
        assert torch.equal(x["instances"].scores_3d, y["instances"].scores_3d) and torch.equal(x["instances"].pred_boxes.tensor, y["instances"].pred_boxes.tensor)


def test_bev_capacity_counts_boxes_not_slots(hiplib):
    """Six nuScenes samples (36 images) on one rank: 36 x 256 detection SLOTS exceed the 8192 entries of the BEV sorter (round 2 refused
    to build this plan), the few hundred actual detections do not.  Against the oracle on identical head maps."""
    from oracle import nuscenes_oracle as N
    from dd3d_amd.synthetic import make_inputs
    from tests.util import bundle
    cfg, sd = bundle("dd3d_nusc_dla34", "dla34_nusc", {"DD3D": {"FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.02}}}})
    model = gpu_model(cfg, sd, use_graph=False)
    B = 36
    inputs = make_inputs(B, 128, 160, dataset="nusc")
    with torch.no_grad():
        ref, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    assert plan.B * plan.det_cap > 8192
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    assert sum(len(r["scores"]) for r in ref) > 100
    for i in range(B):
        _check_final(out[i], ref[i])
        _nusc_check(out[i]["instances"], ref[i], True)


def test_kitti_bev_nms_matches_oracle(hiplib, kitti_dla34):
    """DD3D.INFERENCE.DO_BEV_NMS on the KITTI model (core.py:135-150): per-image BEV rotated NMS between the 2D NMS and the
    resize, poses taken from 'extrinsics' when there is no 'pose'."""
    from dd3d_amd import get_cfg
    from dd3d_amd.structures import Pose
    from dd3d_amd.synthetic import make_inputs
    _, _, sd = kitti_dla34
    cfg = get_cfg("dd3d_kitti_dla34", {"DD3D": {"INFERENCE": {"DO_BEV_NMS": True}, "FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.02}}}})
    model = gpu_model(cfg, sd, use_graph=False)
    assert model.do_bev_nms
    inputs = make_inputs(2, 192, 384, out_hw=(250, 500))
    for i, x in enumerate(inputs):
        x["extrinsics"] = Pose.from_yaw(20.0 * i, (1.0, 2.0 + i, 3.0))
    ref, st = _oracle(cfg, sd, inputs)
    n_nms = sum(len(x["scores"]) for x in st["after_nms"]) if "after_nms" in st else None
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    oracle_heads_to_plan(plan, st, cfg.DD3D.NUM_CLASSES)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    assert sum(len(r["scores"]) for r in ref) < 2 * cfg.DD3D.FCOS2D.INFERENCE.POST_NMS_TOPK  # something got suppressed
    for i in range(2):
        _check_final(out[i], ref[i])


def test_readback_record_equals_the_device_state(hiplib, kitti_dla34):
    """engine.PlanBase.fetch / readback (round 6): the forward's last launch packs detection counts, status word and range-guard maxima
    (dd3d_pack_readback), ONE asynchronous copy into pinned memory follows the forward, `collect` reads nothing else from the device.  The
    record must equal what the device holds, be re-used by further collects of the same forward, and be replaced by the next forward."""
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    model = gpu_model(cfg, sd, use_graph=True)
    a, b = make_inputs(2, 128, 256, seed=5), make_inputs(2, 128, 256, seed=77)
    out_a = model(a)
    plan = model.get_plan(2, 128, 256)
    rb = plan.readback()
    assert rb is plan.readback()  # cached: the requests sharing a forward read one record
    assert rb.status == 0 and torch.equal(rb.counts, plan.det_count.cpu()) and rb.counts.tolist() == [len(o["instances"]) for o in out_a]
    assert torch.equal(rb.amax, plan.amax_values()) and rb.amax.numel() == len(plan.amax_names) and float(rb.amax.max()) > 0
    assert rb.flags.shape == (0, 2)  # (no exchange: no ranks' verdicts)
    out_b = model(b)
    rb2 = plan.readback()
    assert rb2 is not rb and torch.equal(rb2.counts, plan.det_count.cpu()) and rb2.counts.tolist() == [len(o["instances"]) for o in out_b]
    # a forward issued launch by launch WITHOUT the copy (a test replaying by hand): readback packs and fetches on demand
    model.stage_inputs(a, plan=plan)
    plan.launch()
    torch.cuda.synchronize()
    rb3 = plan.readback()
    assert rb3.counts.tolist() == rb.counts.tolist()
    # the status word travels in the record: an overflow raised by collect() comes out of it, with the sample that sizes the next plane scale
    plan.status.fill_(1)
    plan.launch()  # (the stem launch does not clear the status word: it is sticky until read)
    torch.cuda.synchronize()
    from dd3d_amd.engine import HalfRangeOverflow
    with pytest.raises(HalfRangeOverflow) as ei:
        plan.check_status(plan.readback())
    assert ei.value.sampled_max_abs is not None and ei.value.sampled_max_abs > 0 and int(plan.status.cpu()) == 0
