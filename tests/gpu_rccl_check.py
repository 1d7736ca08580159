"""The product transport of the N > 1 step on the one GPU of the test box: a single-rank RCCL ("nccl") group carries the candidate
all_gather between the two captured hipGraph halves (`force_exchange`), device tensors in, device tensors out.  Detections must equal
the single-graph forward of the same model.  (tests/gpu_dist_check.py covers W = 2 ranks over a host-staged gloo gather.)

    python tests/gpu_rccl_check.py [graph|probe]   graph: DD3D_GRAPH_EXCHANGE=1 -- the all_gather captured INSIDE the step's hipGraph
                                                   probe: DD3D_GRAPH_EXCHANGE=probe -- dd3d_amd.parallel.graph_exchange_probe decides
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    one_graph = mode == "graph"
    if mode in ("graph", "probe"):
        os.environ["DD3D_GRAPH_EXCHANGE"] = "1" if one_graph else "probe"
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import DistributedForward, gather_candidates
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    if mode == "probe":  # the capability probe (a tiny captured all_gather on a communicator of its own) decides; both outcomes are legal
        from dd3d_amd.parallel import exchange_selftest, graph_exchange_enabled, graph_exchange_probe
        one_graph = graph_exchange_probe()
        assert graph_exchange_enabled() == one_graph and graph_exchange_probe() == one_graph  # cached, stable
        st = exchange_selftest()
        print(f"probe: captured all_gather usable = {one_graph}; selftest {st['nranks']} rank(s), device {st['devices'][0]}")
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_kitti"))
    model.load_state_dict(sd)
    B, H, W = 2, 192, 384
    inputs = make_inputs(B, H, W, seed=1000)
    ok = True
    for use_graph in (False, True):
        runner = DistributedForward(model, B, H + (-H) % 128, W + (-W) % 128, use_graph=use_graph, force_exchange=True)
        p = runner.plan
        assert p.exchange and p.cand_all.data_ptr() != p.cand.data_ptr()
        assert not use_graph or (runner.step_graph is not None if one_graph else runner.pre_graph is not None)
        p.cand_all.fill_(float("nan"))  # the NMS half must see what RCCL delivered, not stale memory
        out = runner.forward(inputs)
        out = runner.forward(inputs)
        torch.cuda.synchronize()
        ok &= bool(torch.equal(p.cand_all, p.cand) and torch.equal(p.counts_all, p.counts))
        single = build_model(cfg)
        single.load_state_dict(sd)
        ref = single(inputs)
        for o, r in zip(out, ref):
            a, b = o["instances"], r["instances"]
            ok &= len(a) == len(b) and len(a) > 0
            ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
            ok &= torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_boxes3d.quat, b.pred_boxes3d.quat)
    # timing of a whole step of the last runner (captured halves or one captured graph around the collective)
    for _ in range(5):
        runner.step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50):
        runner.step()
    torch.cuda.synchronize()
    print(f"step ({'one graph incl. the all_gather' if one_graph else 'graph half, all_gather, graph half'}): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
    # timing of the exchange itself (ONE all_gather_into_tensor of the rank's record)
    pairs = runner.plan.gather_pairs()
    for _ in range(5):
        gather_candidates(pairs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gather_candidates(pairs)
    e1.record()
    torch.cuda.synchronize()
    print(f"rccl check{' (all_gather inside the graph)' if one_graph else ''}: ok={bool(ok)} exchange {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per step (W=1, {sum(l.numel() * l.element_size() for l, _ in pairs)} B)")
    dist.destroy_process_group()
    assert ok


if __name__ == "__main__":
    main()
