"""The N>1 path's only collective, exercised with world_size 2 on CPU (gloo): rank-major all_gather of the fixed-capacity
candidate buffers + count headers (dd3d_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, F, NS, L, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dd3d_amd.parallel import gather_candidates, init_distributed, owner_of_image
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    cand = torch.randn(B, F, NS, generator=g)
    counts = torch.randint(0, 50, (B, L), generator=g, dtype=torch.int32)
    outsz = torch.full((B, 4), float(rank))
    cand_all, counts_all, outsz_all = torch.zeros(world * B, F, NS), torch.zeros(world * B, L, dtype=torch.int32), torch.zeros(world * B, 4)
    gather_candidates([(cand, cand_all), (counts, counts_all), (outsz, outsz_all)])
    ok = True
    for src in range(world):  # every rank can regenerate every other rank's payload
        g2 = torch.Generator().manual_seed(100 + src)
        c2 = torch.randn(B, F, NS, generator=g2)
        n2 = torch.randint(0, 50, (B, L), generator=g2, dtype=torch.int32)
        ok &= torch.equal(cand_all[src * B:(src + 1) * B], c2) and torch.equal(counts_all[src * B:(src + 1) * B], n2)
        ok &= bool((outsz_all[src * B:(src + 1) * B] == src).all())
        ok &= all(owner_of_image(gi, B) == src for gi in range(src * B, (src + 1) * B))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_candidates_gloo_world2():
    world, B, F, NS, L = 2, 3, 22, 40, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, F, NS, L, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
