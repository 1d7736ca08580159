"""The block-row resolution of the NMS kernels (csrc/postproc.hip::greedy_reduce), restated on the CPU.

Greedy NMS inside one 64-candidate block row is a chain: candidate j is kept iff it is alive and no KEPT earlier candidate of the row
suppresses it.  The kernel does not walk that chain; it iterates   keep <- alive & ~(any(col & keep))   with one 64-lane ballot per
round, where col[j] is the set of earlier rows that suppress j (the column-form diagonal word nms_mask_kernel / bev_mask_kernel write).
This file checks the two facts the kernel relies on: the iteration stops at exactly the sequential answer, and it needs at most
(longest suppression chain + 1) <= 65 rounds.  (The device code itself is covered by tests/test_nms_gpu.py.)
"""
import numpy as np
import pytest


def sequential(sup, alive):
    """sup[i, j] (i < j): i suppresses j.  The torchvision loop restricted to one block row."""
    keep = np.zeros(64, dtype=bool)
    removed = ~alive
    for i in range(64):
        if removed[i]:
            continue
        keep[i] = True
        removed |= sup[i]
    return keep


def ballot_iteration(sup, alive):
    col = sup.T  # col[j, i]: earlier row i suppresses j
    keep = alive.copy()
    rounds = 0
    while True:
        rounds += 1
        nxt = alive & ~(col & keep[None, :]).any(1)
        if (nxt == keep).all():
            return keep, rounds
        keep = nxt


def longest_chain(sup, alive):
    """Longest path i0 -> i1 -> ... of suppression edges among alive candidates (the depth the iteration has to propagate)."""
    depth = np.zeros(64, dtype=int)
    for j in range(64):
        if alive[j]:
            prev = [depth[i] for i in range(j) if alive[i] and sup[i, j]]
            depth[j] = 1 + max(prev, default=0)
    return int(depth.max())


@pytest.mark.parametrize("density", [0.0, 0.02, 0.1, 0.5, 1.0])
def test_ballot_iteration_equals_the_sequential_loop(density):
    rng = np.random.default_rng(int(density * 100))
    for trial in range(200):
        sup = np.triu(rng.random((64, 64)) < density, k=1)
        alive = rng.random(64) < rng.choice([0.3, 0.9, 1.0])
        want = sequential(sup, alive)
        got, rounds = ballot_iteration(sup, alive)
        assert (got == want).all(), (density, trial)
        assert rounds <= longest_chain(sup, alive) + 1 <= 65, (density, trial, rounds)


def test_staircase_needs_as_many_rounds_as_it_is_long():
    sup = np.zeros((64, 64), dtype=bool)
    for i in range(63):
        sup[i, i + 1] = True  # every candidate suppresses the next one only: keep, drop, keep, ...
    alive = np.ones(64, dtype=bool)
    got, rounds = ballot_iteration(sup, alive)
    assert (got == sequential(sup, alive)).all() and got.tolist() == [i % 2 == 0 for i in range(64)]
    assert 32 <= rounds <= 65


# ---- the two index schedules of the NMS kernels, restated with the constants read out of the source
def _constants():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dd3d_amd", "csrc", "postproc.hip")).read()
    val = lambda name: int(eval(re.search(rf"constexpr int {name} = ([0-9*+ ]+);", src).group(1)))  # noqa: E731  (digits, * and + only)
    return {k: val(k) for k in ("PT", "NCAP_MAX", "FIN_SMALL", "FIN_LDS_BYTES", "MASK_BLOCKS")}


def test_tile_walk_covers_the_upper_triangle_once():
    """nms_mask_kernel / bev_mask_kernel: tile t -> (rb, cb) by peeling rows of decreasing length (postproc.hip::tile_of)."""
    def tile_of(t, nwords):
        rb = 0
        while t >= nwords - rb:
            t -= nwords - rb
            rb += 1
        return rb, rb + t
    for nwords in list(range(1, 20)) + [59, 64, 128]:
        tiles = [tile_of(t, nwords) for t in range(nwords * (nwords + 1) // 2)]
        assert sorted(tiles) == [(r, c) for r in range(nwords) for c in range(r, nwords)]


def test_finalize_mask_staging_fits_the_lds_for_every_size():
    """nms_finalize_kernel: the mask words travel to LDS in one piece (n <= FIN_SMALL) or in chunks of R block rows; every chunk
    must fit the stage left beside the index arrays, need <= 16 loads per thread, and make progress."""
    k = _constants()
    PT, FIN_SMALL = k["PT"], k["FIN_SMALL"]
    assert k["FIN_LDS_BYTES"] + 4096 <= 160 * 1024  # dynamic + the kernel's static arrays inside a CU's LDS
    # small mode: removed + kept + sidx + the whole triangle [FIN_SMALL / 64][FIN_SMALL], then the top-k scratch in the same stage
    stage_words = (k["FIN_LDS_BYTES"] - k["NCAP_MAX"] // 64 * 8 - FIN_SMALL * 8) // 8
    assert stage_words >= FIN_SMALL // 64 * FIN_SMALL and stage_words >= FIN_SMALL
    ncap2 = 2048
    while ncap2 <= k["NCAP_MAX"]:
        stage_words = (k["FIN_LDS_BYTES"] - k["NCAP_MAX"] // 64 * 8 - ncap2 * 8) // 8
        assert stage_words >= ncap2  # tkeys | tvals alias the stage after the greedy pass
        for n in range(FIN_SMALL + 1, ncap2 + 1, 61):
            nwords = (n + 63) // 64
            rb0, trips = 0, 0
            while rb0 < nwords:
                wrow = nwords - rb0
                pitch = wrow | 1
                R = min(wrow, stage_words // (64 * pitch), 16 * PT // (64 * wrow))
                assert R >= 1, (ncap2, n, rb0)
                assert R * 64 * pitch <= stage_words and R * 64 * wrow <= 16 * PT
                rb0 += R
                trips += 1
            assert trips <= nwords
        ncap2 *= 2
