"""Parity report of one HIP forward against the CPU oracle (TEST INFRASTRUCTURE: imports oracle/; used by bench.py's cpu_baseline leg,
__graft_entry__.smoke() and the GPU tests -- never by dd3d_amd).

BASELINE.json's metric has two halves: "images/sec ... ; 3D-box L1 vs ref".  `parity_report` is the second half, computed on the image the
bench times (outside the clock): the quantities are the reference's own -- GenericBoxes3D.corners (tridet/structures/boxes3d.py:47-64), the
vectorised box [quat | tvec | size] (boxes3d.py:142-144) -- evaluated on the HIP detections and on the oracle's for the same uint8 image.

  int_mismatches   detections whose integer fields (class, FPN level, location) differ from the oracle's at the same rank -- bar: 0,
                   unless every differing candidate sits ON a selection cut (on_cut_flips, see tests/util.py::candidate_margins) or the
                   difference is a swap of two detections whose oracle scores tie within SWAP_GAP_REL (rank_swaps, rank_swap_gap_rel_max)
  box3d_l1         mean |[tvec | size] difference| over the matched detections (metres), and the same relative to mean |[tvec | size]|
  corners_l1       mean |corner difference| over the 8 corners of the matched boxes (metres), and relative
  depth / size / box2d / score relative errors: the north star's "floats within 1e-3 rel"
"""
import torch

REL_TOL = 1e-3     # north star: box / depth floats within 1e-3 relative
MARGIN_EPS = 2e-6  # |oracle score - cut| of a candidate only one side selected
SWAP_GAP_REL = 1e-4  # two detections may change places in the score-ordered result only over an oracle score gap below this (relative)


def _key(levels, locs, classes):
    return [(int(l), float(x), float(y), int(c)) for l, (x, y), c in zip(levels.tolist(), locs.tolist(), classes.tolist())]


def _rel(a, b, floor=1e-3):
    if b.numel() == 0:
        return 0.0
    den = b.abs().clamp(min=floor * max(float(b.abs().max()), 1e-30))
    return float(((a - b).abs() / den).max())


def parity_report(out, ref, plan=None, stages=None, cfg=None, image=0, ref_image=None):
    """`out`: {"instances": Instances} of the HIP path; `ref`: the oracle's result dict of the same image; `plan` + `stages` (oracle) + `cfg`:
    also count the candidates only one side selected and how far from a selection cut the oracle's score of each sits (`image`: its
    position in the plan, `ref_image`: in the oracle's batch, default the same)."""
    from oracle import dd3d_oracle as O
    o = out["instances"]
    n_hip, n_ref = len(o), len(ref["scores"])
    rep = {"detections_hip": n_hip, "detections_oracle": n_ref}
    ko = _key(o.fpn_levels.cpu(), o.locations.cpu(), o.pred_classes.cpu())
    kr = _key(ref["fpn_levels"], ref["locations"], ref["pred_classes"])
    rep["int_mismatches"] = sum(a != b for a, b in zip(ko, kr)) + abs(n_hip - n_ref)
    # match by (level, location, class): rank order may differ by a swap of two near-equal scores without any field being wrong
    pos = {k: i for i, k in enumerate(kr)}
    pairs = [(i, pos[k]) for i, k in enumerate(ko) if k in pos]
    rep["matched"] = len(pairs)
    # Rank swaps: matched detections that sit at another rank than the oracle's.  The final list is ordered by score, the two sides' scores
    # agree to ~1e-5 relative, so two detections whose ORACLE scores are closer than that may legitimately change places; the report
    # carries the largest oracle score gap such a swap jumped over (bar: SWAP_GAP_REL).
    swaps = [(i, j) for i, j in pairs if i != j]
    rep["rank_swaps"] = len(swaps)
    if swaps:
        key = ref["scores_3d"] if ("scores_3d" in ref and bool((ref["scores_3d"][:-1] >= ref["scores_3d"][1:]).all())) else ref["scores"]
        rep["rank_swap_gap_rel_max"] = max(abs(float(key[min(i, n_ref - 1)]) - float(key[j])) / max(abs(float(key[j])), 1e-30) for i, j in swaps)
    if pairs:
        ih = torch.tensor([p[0] for p in pairs])
        ir = torch.tensor([p[1] for p in pairs])
        has3d = "pred_boxes3d" in ref and o.has("pred_boxes3d")
        rep["box2d_abs_max"] = float((o.pred_boxes.tensor.cpu()[ih] - ref["pred_boxes"][ir]).abs().max())
        rep["box2d_rel_max"] = _rel(o.pred_boxes.tensor.cpu()[ih], ref["pred_boxes"][ir])
        rep["score_rel_max"] = _rel(o.scores.cpu()[ih], ref["scores"][ir])
        if has3d:
            b = {k: v[ir] for k, v in ref["pred_boxes3d"].items()}
            hb = o.pred_boxes3d.to("cpu")
            tv_r = O.boxes3d_tvec(b)
            vec_r = torch.cat([tv_r, b["size"]], 1)
            vec_h = torch.cat([hb.tvec[ih], hb.size[ih]], 1)
            l1 = float((vec_h - vec_r).abs().mean())
            rep["box3d_l1_tvec_size"] = l1
            rep["box3d_l1_tvec_size_rel"] = l1 / max(float(vec_r.abs().mean()), 1e-30)
            c_r = O.boxes3d_corners(b["quat"], tv_r, b["size"])
            c_h = hb.corners[ih]
            cl1 = float((c_h - c_r).abs().mean())
            rep["corners_l1"] = cl1
            rep["corners_l1_rel"] = cl1 / max(float(c_r.abs().mean()), 1e-30)
            rep["depth_rel_max"] = _rel(hb.depth[ih], b["depth"])
            rep["size_rel_max"] = _rel(hb.size[ih], b["size"])
            qa, qb = hb.quat[ih], b["quat"]
            rep["quat_abs_max_up_to_sign"] = float(torch.minimum((qa - qb).abs().amax(1), (qa + qb).abs().amax(1)).max())
            rep["score3d_rel_max"] = _rel(o.scores_3d.cpu()[ih], ref["scores_3d"][ir])
    if plan is not None and stages is not None and cfg is not None:
        from tests.util import candidate_margins
        nh, nr, margins = candidate_margins(plan, stages, cfg, image, ref_image)
        rep["candidates_hip"], rep["candidates_oracle"] = nh, nr
        rep["on_cut_flips"] = sum(m <= MARGIN_EPS for m in margins)
        rep["off_cut_flips"] = sum(m > MARGIN_EPS for m in margins)
        rep["max_flip_margin"] = max(margins) if margins else 0.0
    rep["tolerance_rel"] = REL_TOL
    rep["pass"] = parity_pass(rep)
    return rep


def parity_pass(rep):
    """The bars of the parity tests (tests/test_forward_gpu.py::_check_final), on a report."""
    flips = rep.get("on_cut_flips", 0) + rep.get("off_cut_flips", 0)
    if rep.get("off_cut_flips", 0):
        return False
    if flips == 0:
        # nothing sat on a cut: the same detections with the same integer fields -- at the same ranks, except where two of them tie
        # within the float tolerance of the scores (rank_swaps, each over an oracle score gap <= SWAP_GAP_REL)
        if rep["detections_hip"] != rep["detections_oracle"] or rep.get("matched", 0) != rep["detections_oracle"]:
            return False
        if rep["int_mismatches"] and (rep["int_mismatches"] != rep.get("rank_swaps", 0) or rep.get("rank_swap_gap_rel_max", 0.0) > SWAP_GAP_REL):
            return False
    # The float bars only say something about MATCHED detections: a report in which (nearly) nothing matched must not pass on the
    # absence of numbers (round-5 advisor).  A candidate flipped on a cut can add / remove / displace at most a detection or two.
    if rep.get("matched", 0) < min(rep["detections_hip"], rep["detections_oracle"]) - 2 * flips:
        return False
    ok = True
    for k in ("box3d_l1_tvec_size_rel", "corners_l1_rel", "depth_rel_max", "size_rel_max", "score_rel_max", "score3d_rel_max", "quat_abs_max_up_to_sign",
              "box2d_rel_max"):
        if k in rep:
            ok = ok and rep[k] <= REL_TOL
    return bool(ok)
