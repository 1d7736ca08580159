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
