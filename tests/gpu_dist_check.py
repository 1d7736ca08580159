"""Drive the N > 1 code path (two captured graph halves, candidate all_gather, NMS over all W*B images, rank-offset collect) with
W processes on ONE GPU: gloo carries the gather (staged through the host), everything else is the product path.  Every rank's
detections must equal what a single-rank model returns for the same image, and every rank must hold every image's detections.

    python tests/gpu_dist_check.py [W]
"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import DistributedForward, init_distributed
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    init_distributed(backend="gloo")
    cfg = get_cfg("dd3d_kitti_dla34")
    sd = None
    ok = True
    for use_graph in (False, True):
        model = build_model(cfg)
        sd = sd or make_state_dict(model, calib=load_calib("dla34_kitti"))
        model.load_state_dict(sd)
        B, H, W = 2, 192, 384
        inputs = make_inputs(B, H, W, seed=1000 + rank * B)
        if rank == 1:
            inputs[1]["height"], inputs[1]["width"] = 99, 201
        runner = DistributedForward(model, B, H + (-H) % 128, W + (-W) % 128, use_graph=use_graph)
        out = runner.forward(inputs)
        out = runner.forward(inputs)  # replay
        single = build_model(cfg)
        single.load_state_dict(sd)
        ref = single(inputs)
        for o, r in zip(out, ref):
            a, b = o["instances"], r["instances"]
            ok &= len(a) == len(b) and len(a) > 0 and tuple(a.image_size) == tuple(b.image_size)
            ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
            ok &= torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_boxes3d.quat, b.pred_boxes3d.quat)
        counts = runner.plan.det_count.cpu().tolist()  # a rank finalises its OWN images only
        ok &= len(counts) == B and all(c > 0 for c in counts)
        # ... but the exchange delivered every rank's candidates: the gathered count table holds all W * B images
        gc = runner.plan.gathered_counts().cpu()
        ok &= gc.shape[0] == world * B and bool((gc.sum(1) > 0).all())
        ok &= bool(torch.equal(gc[rank * B:(rank + 1) * B], runner.plan.counts.cpu()))
        dist.barrier()
    # throughput mode: two plan slots, exchange + NMS of one step under the trunk of the next, several steps in flight
    from dd3d_amd.parallel import PipelinedForward
    stream = [make_inputs(B, H, W, seed=50 + 10 * i + rank * B) for i in range(4)]
    refs = [single(x) for x in stream]
    piped = PipelinedForward(model, B, H + (-H) % 128, W + (-W) % 128, depth=2)
    handles = [piped.submit(x) for x in stream[:2]]
    outs = []
    for x in stream[2:]:
        outs.append(piped.result(handles.pop(0)))
        handles.append(piped.submit(x))
    outs += [piped.result(h) for h in handles]
    for out, ref in zip(outs, refs):
        for o, r in zip(out, ref):
            a, b = o["instances"], r["instances"]
            ok &= len(a) == len(b) and len(a) > 0
            ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
            ok &= torch.equal(a.pred_boxes3d.quat, b.pred_boxes3d.quat)
    dist.barrier()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _camera_worker(rank, world, port, ret):
    """NuscenesDD3D with the six cameras of a sample split 3 / 3 over two ranks: the sample's owner (rank 0) must return, for all six
    cameras, what a single rank returns for the whole sample -- the same detections (integer fields exact), attributes, speeds and
    global boxes.  Floats are compared to 1e-5: the two runs convolve batches of 3 and of 6 images, for which the measured tile table
    may pick other tiles / split-K factors (another summation order in the last bits)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import DistributedForward, init_distributed
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    init_distributed(backend="gloo")
    cfg = get_cfg("dd3d_nusc_dla34", {"DD3D": {"FCOS2D": {"INFERENCE": {"PRE_NMS_THRESH": 0.01}}}})  # ~30 detections per camera
    model = build_model(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_nusc"))
    model.load_state_dict(sd)
    H, W, B = 128, 224, 6 // world
    bad = []
    single = build_model(cfg)
    single.load_state_dict(sd)
    # (absolute tolerance relative to the tensor's largest entry: pixel coordinates are O(100), a clipped one may be exactly 0)
    close = lambda x, y: x.shape == y.shape and (x.numel() == 0 or float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max())))
    for use_graph in (False, True):
        runner = DistributedForward(model, B, H + (-H) % 128, W + (-W) % 128, use_graph=use_graph, camera_sharded=True)
        for step in range(2):  # two different samples through the same plan
            sample = make_inputs(6, H, W, dataset="nusc", seed=300 + 20 * step)
            sample[4]["height"], sample[4]["width"] = 99, 201  # one camera is resized on the way out
            out = runner.forward(sample[rank * B:(rank + 1) * B])
            if rank != 0:
                if out != []:
                    bad.append("a rank that owns no sample returned results")
                continue
            ref = single(sample)
            tag = f"graph={use_graph} step={step}"
            if [g for g, _ in out] != list(range(6)) or sum(len(o["instances"]) for _, o in out) < 60:
                bad.append(f"{tag}: images {[g for g, _ in out]} detections {[len(o['instances']) for _, o in out]}")
            for (g, o), r in zip(out, ref):
                a, b = o["instances"], r["instances"]
                if len(a) != len(b) or tuple(a.image_size) != tuple(b.image_size):
                    bad.append(f"{tag} camera {g}: {len(a)} vs {len(b)} detections, size {a.image_size} vs {b.image_size}")
                    continue
                for f in ("pred_classes", "pred_attributes", "fpn_levels", "locations"):
                    if not torch.equal(getattr(a, f), getattr(b, f)):
                        bad.append(f"{tag} camera {g}: {f} differs")
                pairs = [("scores", a.scores, b.scores), ("scores_3d", a.scores_3d, b.scores_3d), ("speeds", a.pred_speeds, b.pred_speeds),
                         ("boxes", a.pred_boxes.tensor, b.pred_boxes.tensor), ("boxes3d", a.pred_boxes3d.vectorize(), b.pred_boxes3d.vectorize()),
                         ("inv_K", a.pred_boxes3d.inv_intrinsics, b.pred_boxes3d.inv_intrinsics),
                         ("global", a.pred_boxes3d_global.vectorize(), b.pred_boxes3d_global.vectorize())]
                for name, x, y in pairs:
                    if not close(x, y):
                        bad.append(f"{tag} camera {g}: {name} max diff {float((x - y).abs().max()) if x.shape == y.shape and x.numel() else -1:.3e}")
        dist.barrier()
    if bad:
        print(f"[rank {rank}]", *bad[:12], sep="\n  ", flush=True)
    ret[rank] = not bad
    dist.destroy_process_group()


def _guard_worker(rank, world, port, ret):
    """The f16x2 range guard across ranks: rank 1's weights push one activation past the half range (same network function: the
    neighbouring norms absorb the factor), rank 0's do not.  The verdict travels in the exchanged records, so BOTH ranks must fall back to
    bf16x3 on the same step -- in the step-by-step runner and in the pipelined one -- and return what a bf16x3 model returns."""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.pop("DD3D_MATH", None)
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import DistributedForward, PipelinedForward, init_distributed
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    init_distributed(backend="gloo")
    cfg = get_cfg("dd3d_kitti_dla34")
    sd = make_state_dict(build_model(cfg), calib=load_calib("dla34_kitti"))
    if rank == 1:
        a, b, f = "backbone.bottom_up.level3.tree1.tree1.conv1.norm", "backbone.bottom_up.level3.tree1.tree1.conv2.norm", 1.0e6  # (beyond 65504 at any plane scale: the staged fallback must end on bf16x3)
        sd = {k: v.clone() for k, v in sd.items()}
        sd[a + ".weight"] *= f
        sd[a + ".bias"] *= f
        sd[b + ".running_mean"] *= f
        sd[b + ".running_var"] *= f * f
    B, H, W = 2, 128, 256
    ref_model = build_model(cfg)
    ref_model.load_state_dict(sd)
    ref_model.math = "bf16x3"
    bad = []

    def same(out, ref, tag):
        for o, r in zip(out, ref):
            x, y = o["instances"], r["instances"]
            if not (len(x) == len(y) > 0 and torch.equal(x.pred_classes, y.pred_classes) and torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor)
                    and torch.equal(x.scores_3d, y.scores_3d)):
                bad.append(f"{tag}: results differ from the bf16x3 model ({len(x)} vs {len(y)} detections)")

    for kind in ("step", "pipelined"):
        model = build_model(cfg)
        model.load_state_dict(sd)
        inputs = [make_inputs(B, H, W, seed=70 + 10 * i + rank * B) for i in range(2)]
        refs = [ref_model(x) for x in inputs]
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            if kind == "step":
                runner = DistributedForward(model, B, H, W, use_graph=True)
                outs = [runner.forward(x) for x in inputs]
            else:
                runner = PipelinedForward(model, B, H, W, depth=2)
                handles = [runner.submit(x) for x in inputs]
                outs = [runner.result(h) for h in handles]
        if not (any("bf16x3" in str(x.message) for x in w) and model.math == "bf16x3"):
            bad.append(f"{kind}: rank {rank} did not fall back (math={model.math}, warnings={[str(x.message)[:60] for x in w]})")
        for i, (o, r) in enumerate(zip(outs, refs)):
            same(o, r, f"{kind} step {i}")
        dist.barrier()
    if bad:
        print(f"[rank {rank}]", *bad[:8], sep="\n  ", flush=True)
    ret[rank] = not bad
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "guard":
        import __graft_entry__ as g
        g.build()
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = mp.Manager().dict()
        mp.spawn(_guard_worker, args=(2, port, ret), nprocs=2, join=True)
        print("range-guard consensus check:", dict(ret))
        assert all(ret.get(r) for r in range(2)), dict(ret)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cameras":
        import __graft_entry__ as g
        g.build()
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = mp.Manager().dict()
        mp.spawn(_camera_worker, args=(2, port, ret), nprocs=2, join=True)
        print("camera-sharded check:", dict(ret))
        assert all(ret.get(r) for r in range(2)), dict(ret)
        return
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import __graft_entry__ as g
    g.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    print("dist check:", dict(ret))
    assert all(ret.get(r) for r in range(world)), dict(ret)


if __name__ == "__main__":
    main()
