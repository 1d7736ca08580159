"""dd3d_nms_finalize called directly (C ABI) against the oracle's restatement of torchvision batched_nms + the fcos2d.py:346-367 top-k.

The forward tests reach the NMS with whatever the synthetic head maps produce; these cases aim at the kernels' own seams instead:
candidate counts around the 64-wide block rows and around the n <= 1024 / n > 1024 and 4n > 4000 (per-class) switches, long
suppression chains inside one block row (the fixed-point resolution needs as many rounds as the chain is long), score ties
(order = concatenated index), several levels with trimmed slots, several images per launch, and the top-k cut with ties.
The kept set and its order must be IDENTICAL to the oracle's.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IDX_FIELD = 8  # a candidate field dd3d_nms_finalize copies verbatim into the detection row: carries the candidate's index here


def _run(levels, thr, post_topk, use_score3d=1, slots=None):
    """levels: per image, a list (one entry per level) of dicts boxes [m,4], score, score3d, cls.  Returns per image the candidate
    indices (position in the level-major concatenation) of the detections, in output order."""
    from dd3d_amd import hip
    lib = hip.lib()
    dev = torch.device("cuda")
    G, L = len(levels), len(levels[0])
    if slots is None:
        topk = max(1, max(len(lv["score"]) for img in levels for lv in img))
        slot_off = [0] * 9
        NS = L * topk
        base = [l * topk for l in range(L)]
    else:
        topk = max(slots)
        base = [int(x) for x in np.cumsum([0] + list(slots))[:-1]]
        NS = int(sum(slots))
        slot_off = base + [NS] + [0] * (9 - L - 1)
    cand = torch.zeros((G, hip.CAND_FIELDS, NS), dtype=torch.float32)
    counts = torch.zeros((G, L), dtype=torch.int32)
    for g, img in enumerate(levels):
        run = 0
        for l, lv in enumerate(img):
            m = len(lv["score"])
            counts[g, l] = m
            sl = slice(base[l], base[l] + m)
            cand[g, 0:4, sl] = lv["boxes"].T
            cand[g, 4, sl], cand[g, 5, sl] = lv["score"], lv["score3d"]
            cand[g, 6, sl] = lv["cls"].to(torch.int32).view(torch.float32)
            cand[g, IDX_FIELD, sl] = torch.arange(run, run + m, dtype=torch.float32)
            run += m
    cand, counts = cand.to(dev), counts.to(dev)
    ncap = (NS + 63) // 64 * 64
    det_cap = max(1, NS)
    w = dict(sort_idx=torch.zeros((G, ncap), dtype=torch.int32, device=dev), sbox=torch.zeros((G, ncap, 4), dtype=torch.float32, device=dev),
             scls=torch.zeros((G, ncap), dtype=torch.int32, device=dev), mask=torch.full((G, ncap, ncap // 64), -1, dtype=torch.int64, device=dev),
             nvalid=torch.zeros((G, 2), dtype=torch.int32, device=dev), det=torch.zeros((G, det_cap, hip.DET_FIELDS), dtype=torch.float32, device=dev),
             det_count=torch.zeros((G, ), dtype=torch.int32, device=dev), out_size=torch.ones((G, 4), dtype=torch.float32, device=dev))
    a = hip.NmsArgs()
    a.cand, a.counts, a.G, a.num_levels, a.topk = cand.data_ptr(), counts.data_ptr(), G, L, topk
    a.do_nms, a.use_score3d, a.nms_thresh, a.post_topk, a.do_postprocess = 1, use_score3d, float(thr), post_topk, 0
    a.out_size, a.sort_idx, a.sbox, a.scls = w["out_size"].data_ptr(), w["sort_idx"].data_ptr(), w["sbox"].data_ptr(), w["scls"].data_ptr()
    a.mask, a.nvalid, a.det, a.det_count, a.det_cap = w["mask"].data_ptr(), w["nvalid"].data_ptr(), w["det"].data_ptr(), w["det_count"].data_ptr(), det_cap
    for i, v in enumerate(slot_off):
        a.slot_off[i] = v
    hip.check(lib.dd3d_nms_finalize(C.byref(a), hip.current_stream()), "nms")
    torch.cuda.synchronize()
    cnt = w["det_count"].cpu().tolist()
    det = w["det"].cpu()
    return [det[g, :cnt[g], IDX_FIELD].to(torch.int64).tolist() for g in range(G)], det, cnt


def _expect(img, thr, post_topk, use_score3d=1):
    from oracle import dd3d_oracle as O
    boxes = torch.cat([lv["boxes"] for lv in img])
    s2, s3, cls = torch.cat([lv["score"] for lv in img]), torch.cat([lv["score3d"] for lv in img]), torch.cat([lv["cls"] for lv in img])
    keep = O.batched_nms(boxes, s3 if use_score3d else s2, cls, thr) if thr > 0 else torch.arange(len(s2))
    n = len(keep)
    if n > post_topk > 0:
        kth, _ = torch.kthvalue(s2[keep], n - post_topk + 1)
        keep = keep[s2[keep] >= kth.item()]
    return keep.tolist()


def _level(gen, m, kind, num_classes=3, tie_levels=0):
    if kind == "clusters":  # a few hundred centres, several jittered boxes on each: heavy suppression, short chains
        k = max(1, m // 6)
        ctr = torch.rand(k, 2, generator=gen) * torch.tensor([1200.0, 360.0])
        pick = torch.randint(0, k, (m, ), generator=gen)
        c = ctr[pick] + torch.randn(m, 2, generator=gen) * 3.0
        wh = 30.0 + 20.0 * torch.rand(m, 2, generator=gen)
    elif kind == "chain":  # a staircase: every box overlaps the next one above the threshold and the one after below it
        c = torch.stack([torch.arange(m, dtype=torch.float32) * 4.0, torch.zeros(m)], 1)
        wh = torch.full((m, 2), 20.0)
    else:  # "sparse": mostly disjoint
        c = torch.rand(m, 2, generator=gen) * torch.tensor([1200.0, 360.0])
        wh = 8.0 + 8.0 * torch.rand(m, 2, generator=gen)
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    if kind == "chain":
        score3d = torch.linspace(0.9, 0.1, m)  # the staircase is visited in order: keep, drop, keep, ...
        cls = torch.zeros(m, dtype=torch.int64)
    else:
        score3d = torch.rand(m, generator=gen) * 0.9 + 0.05
        cls = torch.randint(0, num_classes, (m, ), generator=gen)
    if tie_levels:
        score3d = torch.floor(score3d * tie_levels) / tie_levels + 0.01
    score = torch.rand(m, generator=gen) * 0.9 + 0.05
    if tie_levels:
        score = torch.floor(score * tie_levels) / tie_levels + 0.01
    return dict(boxes=boxes, score=score, score3d=score3d, cls=cls)


SIZES = [0, 1, 2, 63, 64, 65, 127, 129, 500, 1000, 1023, 1024, 1025, 1500, 3000]


@pytest.mark.parametrize("kind", ["clusters", "chain", "sparse"])
def test_single_level_sizes_match_oracle(hiplib, kind):
    gen = torch.Generator().manual_seed(11)
    for n in SIZES:
        img = [_level(gen, n, kind)]
        got, _, _ = _run([img], 0.75 if kind != "chain" else 0.6, 100)
        want = _expect(img, 0.75 if kind != "chain" else 0.6, 100)
        assert got[0] == want, (kind, n, len(got[0]), len(want))
    # the chain really is one: every second box survives, so a block row needs 32 rounds of the fixed-point resolution
    img = [_level(gen, 64, "chain")]
    got, _, _ = _run([img], 0.6, 0)
    assert got[0] == list(range(0, 64, 2))


@pytest.mark.parametrize("n", [200, 1000, 1024, 2500])
def test_score_ties_keep_the_concatenated_order(hiplib, n):
    gen = torch.Generator().manual_seed(n)
    img = [_level(gen, n, "clusters", tie_levels=7)]
    for post_topk in (0, 50):
        got, _, _ = _run([img], 0.5, post_topk)
        assert got[0] == _expect(img, 0.5, post_topk), (n, post_topk)
    got, _, _ = _run([img], 0.5, 50, use_score3d=0)
    assert got[0] == _expect(img, 0.5, 50, use_score3d=0)


def test_levels_with_trimmed_slots_and_several_images(hiplib):
    gen = torch.Generator().manual_seed(3)
    slots = [1000, 1000, 600, 160, 40]  # min(topk, H*W*C) as the engine lays them out
    imgs = []
    for fill in ([1000, 640, 300, 90, 17], [0, 0, 0, 0, 0], [1000, 1000, 600, 160, 40], [3, 0, 64, 0, 1], [400, 350, 200, 60, 14]):
        imgs.append([_level(gen, m, "clusters") for m in fill])
    got, det, cnt = _run(imgs, 0.75, 100, slots=slots)
    for g, img in enumerate(imgs):
        want = _expect(img, 0.75, 100)
        assert got[g] == want, (g, len(got[g]), len(want))
        # the level written to the detection row is the level the candidate came from
        bounds = np.cumsum([0] + [len(lv["score"]) for lv in img])
        lv_of = [int(np.searchsorted(bounds, i, side="right") - 1) for i in want]
        assert det[g, :cnt[g], 7].to(torch.int64).tolist() == lv_of
    # the same images on the dense layout (slot_off all zero: level l at l * topk)
    got_dense, _, _ = _run(imgs, 0.75, 100)
    assert got_dense == got


def test_nms_disabled_and_threshold_zero_pass_everything_in_input_order(hiplib):
    gen = torch.Generator().manual_seed(5)
    img = [_level(gen, 300, "clusters"), _level(gen, 100, "clusters")]
    got, _, _ = _run([img], 0.0, 0)
    assert got[0] == list(range(400))
    got, _, _ = _run([img], 0.0, 100)
    assert got[0] == _expect(img, 0.0, 100)
