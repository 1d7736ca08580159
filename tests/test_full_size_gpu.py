"""BASELINE.json configurations at their stated sizes against the CPU oracle, and the arithmetic modes against their tolerances.

  * configs[2] DD3D-V2-99 KITTI 384x1280, bs=16: head maps of all 16 images <= 1e-4 of their largest entry, candidate flips only ON the
    cut, integer exactness and float parity (<= 1e-3 rel) of the detections on identical head maps;
  * configs[3] architecture, one 6-camera nuScenes sample at 896x1600 on V2-99 (attributes, speeds, global boxes, sample aggregation);
  * the reduced modes the bf16 configurations name (bf16x2, bf16) on DLA-34 at 384x1280 with the tolerance each actually meets
    (measured table: DESIGN.md section 6, tests/gpu_math_modes.py);
  * the half-range guard of the default f16x2 arithmetic: explicit mode raises, default mode falls back to bf16x3.
"""
import warnings

import pytest
import torch

from tests.test_forward_gpu import MARGIN_EPS, REL_TOL, _check_final, _check_head_maps, _nusc_check, _oracle
from tests.util import bundle, candidate_margins, gpu_model, max_abs, oracle_heads_to_plan, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_v99_kitti_bs16_full_size_matches_oracle(hiplib):
    from dd3d_amd.synthetic import make_inputs
    cfg, sd = bundle("dd3d_kitti_v99", "v99_kitti")
    model = gpu_model(cfg, sd, use_graph=False)
    B = 16
    inputs = make_inputs(B, 384, 1280)
    inputs[3]["image"] = inputs[3]["image"][:, :370, :1224].contiguous()  # one raw-KITTI-sized frame in the padded batch
    inputs[3]["height"], inputs[3]["width"] = 370, 1224
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    assert (plan.B, plan.Hp, plan.Wp) == (16, 384, 1280)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    for k, v in st["bottom_up"].items():
        assert max_abs(plan.bottom_up[k].nchw(), v) < 1e-4 * max(1.0, float(v.abs().max())), k
    _check_head_maps(plan, st, C)
    flips = 0
    for i in range(B):
        n_hip, n_ref, margins = candidate_margins(plan, st, cfg, i)
        assert all(m <= MARGIN_EPS for m in margins), (i, n_hip, n_ref, margins)
        flips += len(margins)
    print(f"[margin] V2-99 bs=16: {sum(len(c['scores']) for c in st['candidates'])} oracle candidates, {flips} on-the-cut flips")
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    for i in range(B):
        _check_final(out[i], ref[i])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("policy", ["throughput", "latency"])
def test_dla34_kitti_four_image_plan_matches_oracle(hiplib, policy):
    """The launch plan bench.py's `value` is measured on (round-5 verdict): DD3D-DLA34 384x1280 with FOUR images per launch -- the
    4-image entries of the shipped tile table (the 8-wave 256 x 256 tower tile, the split-K choices of the backbone) -- on four DIFFERENT
    images (one of raw KITTI size inside the padded batch), captured as a hipGraph like a pipeline slot's: backbone features and every head
    map vs the oracle, candidate flips only ON a selection cut, then -- on identical head maps -- integer exactness and <= 1e-3 relative
    floats of the final detections; and end to end through `PipelinedForward(microbatch=4)` (four single-image requests sharing one plan),
    whose results must equal the one-plan forward's bit for bit.  `policy` "latency": the same four images on the tiles of a plan that runs by
    itself (`model(batched_inputs)`: per-launch table, the 192-row tile of the merged FPN output launch) against the same oracle."""
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import make_inputs
    from tests.parity import parity_report
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti")
    model = gpu_model(cfg, sd, use_graph=True)
    model.tile_policy = policy  # "throughput": what a pipeline slot of several requests builds its plan with (engine.tiling.THROUGHPUT_TILE_TABLE): the TIMED plan's tiles
    B = 4
    inputs = [make_inputs(1, 384, 1280, seed=1000 + j)[0] for j in range(B)]  # (the bench's request images)
    inputs[2]["image"] = inputs[2]["image"][:, :370, :1224].contiguous()
    inputs[2]["height"], inputs[2]["width"] = 370, 1224
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    assert (plan.B, plan.Hp, plan.Wp) == (4, 384, 1280) and plan.graph is not None
    towers = [op for op in plan.ops if op.name.startswith("towers.")]
    assert towers and all(op.info["tile_name"] == "256x256w8" for op in towers), [op.info["tile_name"] for op in towers]
    by_name = {op.name: op for op in plan.ops}
    if policy == "throughput":
        assert plan.tile_policy == "throughput" and by_name["level3.tree2.tree1.conv2"].info["tile_name"] == "256x128" and by_name["level5.tree2.conv1"].info["splitk"] == 4
        assert by_name["fpn_outputs"].info["tile_name"] == "256x256w8"
    else:
        assert plan.tile_policy == "latency" and by_name["level3.tree2.tree1.conv2"].info["tile_name"] != "256x128" and by_name["fpn_outputs"].info["tile_name"] == "192x256w8"
    plan.run()
    torch.cuda.synchronize()
    plan.check_status()
    C = cfg.DD3D.NUM_CLASSES
    for k, v in st["bottom_up"].items():
        assert max_abs(plan.bottom_up[k].nchw(), v) < 1e-4 * max(1.0, float(v.abs().max())), k
    _check_head_maps(plan, st, C)
    out_e2e = model.collect(plan, inputs, image_sizes)
    flips = 0
    for i in range(B):
        rep = parity_report(out_e2e[i], ref[i], plan=plan, stages=st, cfg=cfg, image=i)
        assert rep["off_cut_flips"] == 0 and rep["pass"], (i, rep)
        flips += rep["on_cut_flips"]
    print(f"[margin] DLA-34 four-image plan: {sum(len(c['scores']) for c in st['candidates'])} oracle candidates, {flips} on-the-cut flips")
    if policy == "latency":  # (pipeline slots are built with the throughput tiles: other split-K choices, not the same bits)
        oracle_heads_to_plan(plan, st, C)
        plan.launch(first=plan.num_pre_nms_ops - 1)
        torch.cuda.synchronize()
        out = model.collect(plan, inputs, image_sizes)
        for i in range(B):
            _check_final(out[i], ref[i])
        return
    # four single-image requests through one pipeline slot: the same plan geometry, the same detections
    runner = PipelinedForward(model, 1, 384, 1280, depth=2, compute_streams=2, microbatch=4)
    assert runner.plan.tile_policy == "throughput" and [getattr(op, "info", {}).get("tile_name") for op in runner.plan.ops] == [getattr(op, "info", {}).get("tile_name") for op in plan.ops]
    handles = [runner.submit([x]) for x in inputs]  # (the raw-KITTI-sized frame lands on the plan's 384 x 1280 canvas like in the batch above)
    for i, h in enumerate(handles):
        o = runner.result(h)[0]["instances"]
        e = out_e2e[i]["instances"]
        assert tuple(o.image_size) == tuple(e.image_size)
        assert len(o) == len(e) and torch.equal(o.pred_boxes.tensor, e.pred_boxes.tensor) and torch.equal(o.scores_3d, e.scores_3d)
        assert torch.equal(o.pred_classes, e.pred_classes) and torch.equal(o.pred_boxes3d.depth, e.pred_boxes3d.depth)
    # identical head maps -> identical integers, floats within 1e-3
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    for i in range(B):
        _check_final(out[i], ref[i])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["dla34_kitti_384x1280_b1_dets", "v99_kitti_384x1280_b1_dets", "dla34_nusc_896x1600_b6_dets",
                                  "v99_nusc_896x1600_b6_dets"])
def test_full_size_detections_match_the_reference_itself(hiplib, name):
    """BASELINE configs[1] / [2] geometry (384x1280) against detections produced by the reference's own tridet DD3D.forward
    (tests/golden/make_golden.py, run in the build container), end to end from the uint8 image: same detections -- classes, levels,
    locations exactly; boxes / depth / size / scores within 1e-3 relative -- unless a candidate sits ON a selection cut (then the
    difference is bounded by the flips, as in test_forward_matches_oracle; the oracle supplies the margins)."""
    import os

    import numpy as np
    from tests.golden.make_golden import CASES, case_inputs
    from tests.test_forward_gpu import _key
    exp, tag, B, H, W, ragged = CASES[name]
    nusc = "nusc" in exp  # configs[4] geometry: one 6-camera sample, attributes / speeds / global boxes after the sample-level aggregation
    cfg, sd = bundle(exp, tag)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    t = lambda k: torch.from_numpy(g[k])
    inputs = case_inputs(B, H, W, ragged, "nusc" if nusc else "kitti")
    model = gpu_model(cfg, sd, use_graph=True)
    out = model(inputs)
    plan, _ = model.stage_inputs(inputs)
    keys = []
    for i in range(B):
        o = out[i]["instances"]
        keys.append((_key(o.fpn_levels.cpu(), o.locations.cpu(), o.pred_classes.cpu()), _key(t(f"det{i}_levels"), t(f"det{i}_locations"), t(f"det{i}_classes"))))
    st = None
    if any(ko != kr for ko, kr in keys):  # the oracle is only needed to show that a differing candidate sits ON a cut
        if nusc:
            from oracle import nuscenes_oracle as N
            with torch.no_grad():
                _, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
        else:
            _, st = _oracle(cfg, sd, inputs)
    for i in range(B):
        margins = []
        if st is not None:
            _, _, margins = candidate_margins(plan, st, cfg, i)
            assert all(m <= MARGIN_EPS for m in margins), margins
        o = out[i]["instances"]
        ko, kr = keys[i]
        assert len(kr) > 20
        if not margins:
            assert ko == kr, (len(ko), len(kr))  # same detections in the same (score_3d) order
        else:
            assert len(set(ko) ^ set(kr)) <= 4 * len(margins), (len(ko), len(kr), len(margins))
        common = [k for k in kr if k in set(ko)]
        io, ir = [ko.index(k) for k in common], [kr.index(k) for k in common]
        assert max_abs(o.pred_boxes.tensor[io], t(f"det{i}_boxes")[ir]) < REL_TOL * float(t(f"det{i}_boxes").abs().max())
        b3 = o.pred_boxes3d
        assert rel_err(b3.depth[io], t(f"det{i}_depth")[ir]) < REL_TOL and rel_err(b3.size[io], t(f"det{i}_size")[ir]) < REL_TOL
        assert rel_err(o.scores_3d[io], t(f"det{i}_scores_3d")[ir]) < REL_TOL and rel_err(o.scores[io], t(f"det{i}_scores")[ir]) < REL_TOL
        assert max_abs(b3.proj_ctr[io], t(f"det{i}_proj_ctr")[ir]) < REL_TOL * float(t(f"det{i}_proj_ctr").abs().max())
        gq = t(f"det{i}_quat")[ir]
        assert float(torch.minimum((b3.quat[io].cpu() - gq).abs().amax(1), (b3.quat[io].cpu() + gq).abs().amax(1)).max()) < REL_TOL
        assert max_abs(b3.tvec[io], t(f"det{i}_tvec")[ir]) < REL_TOL * float(t(f"det{i}_tvec").abs().max())
        if nusc:
            assert torch.equal(o.pred_attributes.cpu()[io], t(f"det{i}_attributes")[ir])
            assert rel_err(o.pred_speeds[io], t(f"det{i}_speeds")[ir]) < REL_TOL
            gl, gg = o.pred_boxes3d_global.vectorize().cpu()[io], t(f"det{i}_global")[ir]
            assert max_abs(gl[:, 4:], gg[:, 4:]) < REL_TOL * float(gg[:, 4:].abs().max())  # global tvec (~1e3 m) and size
            assert float(torch.minimum((gl[:, :4] - gg[:, :4]).abs().amax(1), (gl[:, :4] + gg[:, :4]).abs().amax(1)).max()) < REL_TOL
        print(f"[golden] {name}: image {i}: {len(kr)} reference detections, {len(common)} shared, {len(margins)} on-the-cut flips")


@pytest.mark.timeout(900)
def test_dla34_nuscenes_batch_full_size_matches_oracle(hiplib):
    """BASELINE.json configs[4] geometry: DD3D-DLA34 on nuScenes, one 6-camera sample at 896x1664 (900x1600 -> ResizeShortestEdge 896
    -> 896x1593, padded to the /128 canvas of the P6/P7 FPN, dla.py:559) -- head maps, attributes / speeds, sample aggregation."""
    from dd3d_amd.synthetic import make_inputs
    from oracle import nuscenes_oracle as N
    cfg, sd = bundle("dd3d_nusc_dla34", "dla34_nusc")
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(6, 896, 1600, dataset="nusc")
    for x in inputs:
        x["image"] = x["image"][:, :, :1593].contiguous()
        x["height"], x["width"] = 900, 1600
    with torch.no_grad():
        ref, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    assert (plan.B, plan.Hp, plan.Wp) == (6, 896, 1664)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    _check_head_maps(plan, st, C)
    flips = 0
    for i in range(6):
        _, _, margins = candidate_margins(plan, st, cfg, i)
        assert all(m <= MARGIN_EPS for m in margins), (i, margins)
        flips += len(margins)
    print(f"[margin] DLA-34 nuScenes 896x1664 b6: {sum(len(c['scores']) for c in st['candidates'])} oracle candidates, {flips} on-the-cut flips")
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    assert sum(len(r["scores"]) for r in ref) > 0
    for i in range(6):
        _check_final(out[i], ref[i])
        _nusc_check(out[i]["instances"], ref[i], True)


@pytest.mark.timeout(900)
def test_v99_nuscenes_sample_full_size_matches_oracle(hiplib):
    from dd3d_amd.synthetic import make_inputs
    from oracle import nuscenes_oracle as N
    cfg, sd = bundle("dd3d_nusc_v99", "v99_nusc")
    model = gpu_model(cfg, sd, use_graph=False)
    inputs = make_inputs(6, 896, 1600, dataset="nusc")
    for x in inputs:  # ResizeShortestEdge(896) of a 900x1600 frame is 896x1593; detections are reported at the raw size
        x["image"] = x["image"][:, :, :1593].contiguous()
        x["height"], x["width"] = 900, 1600
    with torch.no_grad():
        ref, st = N.nuscenes_dd3d_forward(sd, cfg, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    assert (plan.Hp, plan.Wp) == (896, 1600)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    _check_head_maps(plan, st, C)
    oracle_heads_to_plan(plan, st, C)
    plan.launch(first=plan.num_pre_nms_ops - 1)
    torch.cuda.synchronize()
    out = model.collect(plan, inputs, image_sizes)
    assert sum(len(r["scores"]) for r in ref) > 0
    for i in range(6):
        _check_final(out[i], ref[i])
        _nusc_check(out[i]["instances"], ref[i], True)


# what each mode meets on DLA-34 at 384x1280 (measured: profiles/r02_math_modes.json): head maps relative to their largest entry,
# decoded floats relative
MODE_TOL = {"f16x2": (1e-4, REL_TOL), "bf16x3": (1e-4, REL_TOL), "bf16x2": (4e-4, REL_TOL), "bf16": (8e-2, 1e-1)}


@pytest.mark.parametrize("mode", list(MODE_TOL))
def test_arithmetic_modes_meet_their_tolerances(hiplib, kitti_dla34, mode):
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    model = gpu_model(cfg, sd, use_graph=False, math=mode)
    inputs = make_inputs(1, 384, 1280)
    ref, st = _oracle(cfg, sd, inputs)
    plan, image_sizes = model.stage_inputs(inputs)
    plan.run()
    torch.cuda.synchronize()
    C = cfg.DD3D.NUM_CLASSES
    map_tol, dec_tol = MODE_TOL[mode]
    for l in range(len(st["logits"])):
        for name, got, want in [("logits", plan.cls_maps[l].nchw(0, C), st["logits"][l]), ("box2d_reg", plan.b2d_maps[l].nchw(0, 4), st["box2d_reg"][l]),
                                ("centerness", plan.b2d_maps[l].nchw(4, 1), st["centerness"][l]), ("depth", plan.b3d_maps[l].nchw(6 * C, C), st["depth"][l])]:
            assert max_abs(got, want) <= map_tol * max(1.0, float(want.abs().max())), (mode, name, l)
    out = model.collect(plan, inputs, image_sizes)[0]["instances"]
    r = ref[0]
    key = lambda lv, loc, cl: [(int(a), float(x), float(y), int(c)) for a, (x, y), c in zip(lv.tolist(), loc.tolist(), cl.tolist())]
    ko, kr = key(out.fpn_levels.cpu(), out.locations.cpu(), out.pred_classes.cpu()), key(r["fpn_levels"], r["locations"], r["pred_classes"])
    common = [k for k in ko if k in set(kr)]
    assert len(common) >= (0.8 if mode == "bf16" else 0.98) * len(kr), (mode, len(ko), len(kr), len(common))
    io, ir = [ko.index(k) for k in common], [kr.index(k) for k in common]
    assert max_abs(out.pred_boxes.tensor[io], r["pred_boxes"][ir]) <= dec_tol * max(1.0, float(r["pred_boxes"].abs().max()))
    assert rel_err(out.pred_boxes3d.depth[io], r["pred_boxes3d"]["depth"][ir]) <= dec_tol
    assert rel_err(out.pred_boxes3d.size[io], r["pred_boxes3d"]["size"][ir]) <= dec_tol
    if mode in ("f16x2", "bf16x3"):  # the f32-equivalent modes: flips only ON the cut
        _, _, margins = candidate_margins(plan, st, cfg, 0)
        assert all(m <= MARGIN_EPS for m in margins), margins


def test_half_range_guard_of_the_default_arithmetic(hiplib, kitti_dla34, monkeypatch):
    """DD3D_MATH_F16X2 holds activations as IEEE halves of value * plane scale.  With an absurd plane scale every activation overflows:
    the kernels flag it, an explicitly requested f16x2 forward raises, and a model on the default arithmetic switches to the three-term
    bf16 split (with a warning) and still agrees with the oracle."""
    from dd3d_amd import hip
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    monkeypatch.setenv("DD3D_F16_ACT_SCALE", str(2**22))
    monkeypatch.delenv("DD3D_MATH", raising=False)
    inputs = make_inputs(1, 128, 256)
    explicit = gpu_model(cfg, sd, use_graph=False, math="f16x2")
    with pytest.raises(FloatingPointError, match="half range"):
        explicit(inputs)
    model = gpu_model(cfg, sd, use_graph=True, math=None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = model(inputs)
    assert any("bf16x3" in str(x.message) for x in w) and model.math == "bf16x3"
    assert next(iter(model._plans.values())).math == hip.MATH_BF16X3
    ref, _ = _oracle(cfg, sd, inputs)
    assert len(out[0]["instances"]) == len(ref[0]["scores"]) > 0
    assert rel_err(out[0]["instances"].pred_boxes3d.depth, ref[0]["pred_boxes3d"]["depth"]) < REL_TOL


def _scaled_between(sd, bn_a, bn_b, factor):
    """The same network function with the activation BETWEEN norm `bn_a` (+ ReLU + conv) and norm `bn_b` multiplied by `factor`:
    bn_a's affine is scaled, bn_b's running statistics absorb it (ReLU and the convolution are positively homogeneous)."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd[bn_a + ".weight"] *= factor
    sd[bn_a + ".bias"] *= factor
    sd[bn_b + ".running_mean"] *= factor
    sd[bn_b + ".running_var"] *= factor * factor
    return sd


def test_half_range_guard_with_realistic_statistics(hiplib, kitti_dla34, monkeypatch):
    """Round-2 verdict: the range of the default f16x2 arithmetic exercised by a NETWORK whose statistics move, not by an absurd plane
    scale.  One mid-network activation (DLA level3, between the two norms of a BasicBlock) is scaled so that its largest entry is
      ~2000+  (inside the range, 4094 at plane scale 16)  -> no flag, parity with the oracle on the same weights;
      ~6000   (outside)                                   -> the overflow bit: explicit f16x2 raises BEFORE results are returned, the
                                                            default arithmetic lowers its plane scale to 4 (range 16376), STAYS on f16x2 and
                                                            agrees with the oracle; ~40000 -> plane scale 1; ~2e5 -> bf16x3;
      ~1e-6   (far below the pair's absolute floor's useful range) -> the per-tensor maximum trips the underflow side of the guard:
                                                            explicit f16x2 raises, the default falls back and agrees with the oracle."""
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    monkeypatch.delenv("DD3D_MATH", raising=False)
    monkeypatch.delenv("DD3D_F16_ACT_SCALE", raising=False)
    inputs = make_inputs(1, 128, 256)
    bn_a, bn_b = "backbone.bottom_up.level3.tree1.tree1.conv1.norm", "backbone.bottom_up.level3.tree1.tree1.conv2.norm"
    assert bn_a + ".running_var" in sd and bn_b + ".running_var" in sd
    base = gpu_model(cfg, sd, use_graph=False, math="f16x2")
    base(inputs)
    plan = next(iter(base._plans.values()))
    slot = plan.amax_names.index("level3.tree1.tree1.conv1")
    a0 = float(plan.amax_values()[slot]) / plan.act_scale  # largest (sampled) activation between the two norms
    every = plan.amax_values() / plan.act_scale
    print(f"[range] max |activation| per plane-writing conv: min {float(every.min()):.3g} max {float(every.max()):.3g}; level3.tree1.tree1.conv1 {a0:.3g}")
    assert a0 > 0 and float(every.min()) > plan.AMAX_FLOOR / plan.act_scale  # the synthetic network itself sits inside the range
    C = cfg.DD3D.NUM_CLASSES

    def check(model, sd_x):
        ref, st = _oracle(cfg, sd_x, inputs)
        out = model(inputs)
        p = next(iter(model._plans.values()))
        _check_head_maps(p, st, C)
        assert len(out[0]["instances"]) == len(ref[0]["scores"]) > 0

    # ~2000 on the guard's sample of the outputs (a lower bound of the true maximum, which must stay below 4094): inside
    sd_in = _scaled_between(sd, bn_a, bn_b, 2000.0 / a0)
    m = gpu_model(cfg, sd_in, use_graph=False, math="f16x2")
    check(m, sd_in)
    p = next(iter(m._plans.values()))
    assert 1800.0 < float(p.amax_values()[slot]) / p.act_scale < 2200.0
    # Beyond 4094 the explicit mode raises; the DEFAULT arithmetic first widens the half range -- plane scale 16 -> 4 -> 1, i.e. activations up to
    # 16376 -> 65504, still f16x2 (engine.plan.relax_arithmetic) -- and only then, or on an underflow, moves to bf16x3; every outcome agrees with
    # the oracle on the same weights.   (target, error text, plane scale the default ends on or None = bf16x3)
    for target, what, end_scale in ((6000.0, "half range", 4.0), (40000.0, "half range", 1.0), (2.0e5, "half range", None), (1e-6, "useful part", None)):
        sd_x = _scaled_between(sd, bn_a, bn_b, target / a0)
        with pytest.raises(FloatingPointError, match=what):
            gpu_model(cfg, sd_x, use_graph=False, math="f16x2")(inputs)
        dflt = gpu_model(cfg, sd_x, use_graph=True, math=None)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            check(dflt, sd_x)
        msgs = [str(x.message) for x in w]
        p = next(iter(dflt._plans.values()))
        if end_scale is None:
            assert any("bf16x3" in m for m in msgs) and dflt.math == "bf16x3", (target, msgs)
        else:
            assert dflt.math is None and dflt.act_scale == end_scale and p.act_scale == end_scale, (target, dflt.math, dflt.act_scale)
            assert any("plane scale lowered" in m for m in msgs) and not any("switching this model" in m for m in msgs), msgs
            assert 0.8 * target < float(p.amax_values()[slot]) / p.act_scale < 1.2 * target  # the activation is really there, inside the wider range


def _fpn_outlier(sd, level, channel, factor):
    """The state dict with ONE output channel of an FPN lateral scaled by `factor` (its FrozenBN affine: weight and bias) -- a heavy-tailed
    pyramid: one channel of every map from that level down carries `factor` times the others' magnitude."""
    sd = {k: v.clone() for k, v in sd.items()}
    for k in ("weight", "bias"):
        sd[f"backbone.fpn_lateral{level}.norm.{k}"][channel] *= factor
    return sd


@pytest.mark.timeout(900)
def test_heavy_tailed_fpn_statistics_at_full_size(hiplib, kitti_dla34, monkeypatch):
    """Round-4 verdict item 9: the f16x2 default on activation statistics a real checkpoint may have and the synthetic weights do not -- a
    50x outlier channel in the FPN (384 x 1280).  The forward stays on f16x2, agrees with the oracle on the same weights, and REPORTS how
    near the range guard it came (PlanBase.range_headroom, the warning below DD3D_RANGE_WARN_X); a 3000x outlier leaves the half range at the default
    plane scale: the default arithmetic lowers the scale and stays on f16x2; a 60000x outlier leaves every half range: bf16x3; both agree."""
    from dd3d_amd.synthetic import make_inputs
    cfg, _, sd = kitti_dla34
    monkeypatch.delenv("DD3D_MATH", raising=False)
    monkeypatch.delenv("DD3D_F16_ACT_SCALE", raising=False)
    inputs = make_inputs(1, 384, 1280)
    C = cfg.DD3D.NUM_CLASSES
    base = gpu_model(cfg, sd, use_graph=True)
    base(inputs)
    h0 = next(iter(base._plans.values())).range_headroom()
    assert h0["overflow_headroom_x"] > 100 and h0["underflow_headroom_x"] > 100  # the synthetic network: two decades from either end

    sd50 = _fpn_outlier(sd, 4, 5, 50.0)
    _, st = _oracle(cfg, sd50, inputs)
    m = gpu_model(cfg, sd50, use_graph=True)
    monkeypatch.setenv("DD3D_RANGE_WARN_X", "1e9")  # (any headroom is "low": the warning's text is checked, not its threshold)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = m(inputs)
    monkeypatch.delenv("DD3D_RANGE_WARN_X")
    assert m.math is None and any("range headroom" in str(x.message) for x in w)
    p = next(iter(m._plans.values()))
    h = p.range_headroom()
    print(f"[range] 50x FPN outlier channel: largest activation {h['largest_activation']:.4g} in {h['largest_in']} "
          f"(overflow at {h['overflow_at']:g}: headroom {h['overflow_headroom_x']:.1f}x; baseline {h0['overflow_headroom_x']:.0f}x)")
    assert 1.0 < h["overflow_headroom_x"] < h0["overflow_headroom_x"] / 5  # the outlier shows in the report, and the guard did not trip
    _check_head_maps(p, st, C)
    _, _, margins = candidate_margins(p, st, cfg, 0)
    assert all(mg <= MARGIN_EPS for mg in margins), margins
    assert len(out[0]["instances"]) > 0

    # x 3000: ~6000 leaves the half range at plane scale 16 -- explicit f16x2 raises; the default lowers the plane scale to 4 and stays on f16x2
    sd3k = _fpn_outlier(sd, 4, 5, 3000.0)
    _, st3 = _oracle(cfg, sd3k, inputs)
    with pytest.raises(FloatingPointError, match="half range"):
        gpu_model(cfg, sd3k, use_graph=False, math="f16x2")(inputs)
    d = gpu_model(cfg, sd3k, use_graph=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d(inputs)
    assert d.math is None and d.act_scale == 4.0 and any("plane scale lowered" in str(x.message) for x in w)
    _check_head_maps(next(iter(d._plans.values())), st3, C)
    # x 60000: beyond 65504 whatever the plane scale -- the default ends on bf16x3 and still agrees
    sd60k = _fpn_outlier(sd, 4, 5, 60000.0)
    _, st60 = _oracle(cfg, sd60k, inputs)
    e = gpu_model(cfg, sd60k, use_graph=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        e(inputs)
    assert any("bf16x3" in str(x.message) for x in w) and e.math == "bf16x3"
    _check_head_maps(next(iter(e._plans.values())), st60, C)
