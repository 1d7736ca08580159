"""PipelinedForward (two plan slots, exchange + NMS of step i under the trunk of step i+1) must return, for a stream of different
inputs, exactly what the one-step-at-a-time forward returns; with `nccl` the exchange goes through a single-rank RCCL group.

    python tests/gpu_pipeline_check.py [nccl | streams | microbatch]
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def microbatch_main():
    """Slots whose launch plan covers M = 3 single-image requests (two slots): seven requests of different images -- the last slot is only
    partly filled and is flushed by result() -- must come back exactly as the one-at-a-time forward returns them, whatever shares their slot;
    then the bench loop (steps on resident inputs + flush) and a slot re-used after its results were read."""
    import torch
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    torch.cuda.set_device(0)
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    H, W, M = 192, 384, 3
    stream = [make_inputs(1, H, W, seed=10 + 7 * i) for i in range(7)]
    stream[3][0]["height"], stream[3][0]["width"] = 99, 201
    single = build_model(cfg)
    single.load_state_dict(model.state_dict())
    # the reference result of a request: the same image through a plan of the SAME batch size (the measured tile table may choose another
    # split-K for another batch size, which moves the last bits), alone in it
    ref = []
    for x in stream:
        out = single(x * M)
        ref.append(out[:1])
    runner = PipelinedForward(model, 1, H + (-H) % 128, W + (-W) % 128, depth=2, compute_streams=2, microbatch=M)
    ok = True
    handles = [runner.submit(x) for x in stream[:6]]  # two full slots in flight
    outs = [runner.result(h) for h in handles[:3]]     # first slot read -> it may be re-used
    handles.append(runner.submit(stream[6]))           # goes to slot 0, position 0; the slot stays partly filled
    outs += [runner.result(h) for h in handles[3:]]    # the last result() flushes the partial slot
    n_det = 0
    for out, r in zip(outs, ref):
        a, b = out[0]["instances"], r[0]["instances"]
        n_det += len(a)
        ok &= len(out) == 1 and len(a) == len(b) and len(a) > 0 and tuple(a.image_size) == tuple(b.image_size)
        ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
        ok &= torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_boxes3d.quat, b.pred_boxes3d.quat)
        ok &= torch.equal(a.pred_boxes3d.inv_intrinsics, b.pred_boxes3d.inv_intrinsics)
    # the bench loop: 8 steps = 2 full slots + 2 requests in a third run, flushed
    runner.stage_all(stream[0])
    for _ in range(8):
        runner.step()
    runner.synchronize()
    for slot in runner.slots:
        slot.requests[0] = (stream[0], [(H, W)])
        last = runner.result((slot, 0, slot.generation))
        ok &= torch.equal(last[0]["instances"].scores_3d, ref[0][0]["instances"].scores_3d)
    print(f"pipeline check (micro-batched slots, {M} requests per slot, 2 slots): ok={bool(ok)} detections={n_det}")
    assert ok


def fallback_main():
    """The range guard inside the pipeline: with an absurd plane scale every activation of the default f16x2 arithmetic overflows; a
    single-rank runner on the default arithmetic must drain, rebuild its slots on bf16x3, re-run the requests in flight (here: one full
    slot and one partly filled slot) and return what a bf16x3 model returns."""
    import warnings
    import torch
    os.environ["DD3D_F16_ACT_SCALE"] = str(2**22)
    os.environ.pop("DD3D_MATH", None)
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    torch.cuda.set_device(0)
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    H, W, M = 128, 256, 2
    stream = [make_inputs(1, H, W, seed=40 + 3 * i) for i in range(3)]
    ref_model = build_model(cfg)
    ref_model.load_state_dict(model.state_dict())
    ref_model.math = "bf16x3"
    ref = [ref_model(x * M)[:1] for x in stream]
    runner = PipelinedForward(model, 1, H, W, depth=2, compute_streams=2, microbatch=M)
    handles = [runner.submit(x) for x in stream]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        outs = [runner.result(h) for h in handles]
    ok = any("bf16x3" in str(x.message) for x in w) and model.math == "bf16x3"
    for out, r in zip(outs, ref):
        a, b = out[0]["instances"], r[0]["instances"]
        ok &= len(a) == len(b) and len(a) > 0 and torch.equal(a.pred_classes, b.pred_classes)
        ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
    # ... and the rebuilt pipeline keeps serving
    more = runner.result(runner.submit(stream[0]))
    ok &= torch.equal(more[0]["instances"].scores_3d, ref[0][0]["instances"].scores_3d)
    print(f"pipeline check (range guard -> bf16x3 fallback inside the runner): ok={bool(ok)}")
    assert ok


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    if mode == "microbatch":
        return microbatch_main()
    if mode == "fallback":
        return fallback_main()
    use_nccl = mode == "nccl"
    depth, streams = (4, 4) if mode == "streams" else (2, 1)  # "streams": the bench default, consecutive steps share the chip
    from dd3d_amd import build_model, get_cfg
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    torch.cuda.set_device(0)
    if use_nccl:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_kitti"))
    model.load_state_dict(sd)
    B, H, W = 2, 192, 384
    stream = [make_inputs(B, H, W, seed=10 + 7 * i) for i in range(7)]
    stream[3][1]["height"], stream[3][1]["width"] = 99, 201
    ref = [model(x) for x in stream]
    runner = PipelinedForward(model, B, H + (-H) % 128, W + (-W) % 128, depth=depth, force_exchange=use_nccl, compute_streams=streams)
    ok = True
    # `depth` steps in flight before the first result is read; then interleaved
    handles = [runner.submit(x) for x in stream[:depth]]
    outs = []
    for i in range(depth, len(stream)):
        outs.append(runner.result(handles.pop(0)))
        handles.append(runner.submit(stream[i]))
    outs += [runner.result(h) for h in handles]
    n_det = 0
    for out, r in zip(outs, ref):
        for o, q in zip(out, r):
            a, b = o["instances"], q["instances"]
            n_det += len(a)
            ok &= len(a) == len(b) and len(a) > 0 and tuple(a.image_size) == tuple(b.image_size)
            ok &= torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores_3d, b.scores_3d)
            ok &= torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_boxes3d.quat, b.pred_boxes3d.quat)
            ok &= torch.equal(a.pred_boxes3d.inv_intrinsics, b.pred_boxes3d.inv_intrinsics)
    # the bench loop: steps on resident inputs, back to back
    runner.stage_all(stream[0])
    for _ in range(20):
        slot = runner.step()
    torch.cuda.synchronize()
    slot.requests[0] = (stream[0], [(H, W)] * B)  # (step() runs on resident inputs: name them for collect())
    last = runner.result(slot)
    ok &= all(torch.equal(o["instances"].scores_3d, q["instances"].scores_3d) for o, q in zip(last, ref[0]))
    print(f"pipeline check ({'nccl exchange' if use_nccl else 'single rank'}, {depth} slots / {streams} compute streams): ok={bool(ok)} detections={n_det}")
    if use_nccl:
        dist.destroy_process_group()
    assert ok


if __name__ == "__main__":
    main()
