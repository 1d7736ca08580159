#!/usr/bin/env python
"""Throughput benchmark of the DD3D-DLA34 forward path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one forward pass of the hot path (uint8 image already resident in HBM -> final detections in HBM) over one
batch of synthetic 384x1280 KITTI-shaped frames, ``--batch`` images per GPU (default 1 = BASELINE.json configs[1]
"DD3D-DLA34 KITTI3D 384x1280 bs=1 fp32 inference"; one image per GPU per step as the north star shards them).
For N > 1 every rank forwards its own images and the step includes ONE RCCL all_gather of every rank's decoded-candidate record, after
which each rank runs the batched NMS of the images it owns (dd3d_amd/parallel.py).  Steps are issued through PipelinedForward: `--pipeline`
plan slots on `--compute-streams` compute streams + one exchange/NMS stream; a slot's launch plan covers `--microbatch` queued
single-image requests (a step = one request; the slot's graphs are enqueued when its last request arrives, the collective runs under
the next slots' trunks).  `--pipeline 0` issues one step at a time and that figure is also reported in `config`.  Every one of the K
timed steps does all of its work and is complete before the closing synchronize (a partly filled slot is flushed and runs in full).
Every slot's graphs are replayed once at construction and the untimed warm-up covers every slot at least twice, whatever --warmup says.

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel (the head-tower implicit-GEMM launch of the slot's plan:
conv_igemm_planes_row_kernel<4,2,2,4,2,4,false,2> of csrc/conv_planes_row.hip at the default four images per launch): algorithmic FLOPs of
one launch / its mean duration measured here with HIP events on the launch stream (``traffic``: the PMC-derived HBM bytes of that launch
geometry, profiles/r06_tower_hbm_bytes.json; ``measured_mfma_ceiling_on_real_operands_tflops``: tests/tools' dd3d_tools_mfma_probe timed in this run), against the MFMA roofline of the arithmetic in use -- 2500 TFLOP/s dense 16-bit MFMA divided by the
matrix products spent per f32 product (``--math``: f16x2 3, bf16x3 6, bf16x2 3, bf16 1; the f32-input MFMA peak 157.3 TFLOP/s for f32).
``blocks`` repeats the timed block a few times so that a reader can tell box / clock variance from a regression.  ``cpu_baseline`` is
the CPU oracle (a restatement "port" of the reference forward) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# One hardware queue per stream of the default issue mode (5 compute streams + the exchange / NMS stream) instead of the runtime's 4: the
# streams then never share a queue.  Measured on one box: 1372 img/s against 1355 with the default, 1344 with 8 (profiles/r03k_hw_queues.txt).
# Must be in the environment before the HIP runtime starts; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(os.environ.get("DD3D_BENCH_COMPUTE_STREAMS", "5")) + 1))
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails with `hipIpcGetMemHandle: invalid argument`); the driver exports it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense
# Split-operand arithmetic spends several 16-bit MFMA products per f32 product; its roofline in f32-equivalent FLOP/s is the
# dense 16-bit peak divided by that count (the f16 and bf16 pipes run at the same rate)
PRODUCTS = {"f16x2": 3, "bf16x3": 6, "bf16x2": 3, "bf16": 1}
DTYPES = {
    "f16x2": "f32 (f16x2 split-operand MFMA: every f32 operand = two IEEE-half terms, 3 cross products, f32 accumulate; error vs a float64 "
             "convolution equals the exact-f32 MFMA kernel's, tests/gpu_math_modes.py)",
    "bf16x3": "f32 (bf16x3 split-operand MFMA: every f32 operand = three bf16 terms, 6 cross products, f32 accumulate)",
    "bf16x2": "bf16x2 (two bf16 terms per operand, 3 cross products, f32 accumulate; ~1e-5 relative -- a reduced mode)",
    "bf16": "bf16 (operands rounded to bf16, f32 accumulate -- a reduced mode)",
    "f32": "f32",
}
GFLOP_PER_IMAGE = 220.77  # DD3D-DLA34 KITTI 384x1280, 2 x 110.384 GMAC


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying the hipGraph")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("DD3D_BENCH_PIPELINE", "5")),
                    help="plan slots of dd3d_amd.parallel.PipelinedForward (exchange + NMS of step i overlap the trunk of step i+1); "
                         "0 = one step at a time")
    ap.add_argument("--compute-streams", type=int, default=int(os.environ.get("DD3D_BENCH_COMPUTE_STREAMS", "5")),
                    help="PipelinedForward: streams the slots' trunks are issued on (> 1 lets consecutive steps share the chip)")
    ap.add_argument("--microbatch", type=int, default=int(os.environ.get("DD3D_BENCH_MICROBATCH", "4")),
                    help="PipelinedForward: queued requests (steps) one slot's launch plan covers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=200, help="upper bound; the CPU leg stops after ~12 s of work")
    ap.add_argument("--math", default=None, help="arithmetic of the convolutions: f16x2 (default) | bf16x3 | f32 | bf16x2 | bf16 (dd3d_amd.engine.default_math)")
    ap.add_argument("--repeat-blocks", type=int, default=5, help="extra timed blocks of --steps steps each (median / min / max reported in `blocks`)")
    ap.add_argument("--e2e-requests", type=int, default=60,
                    help="requests of the end-to-end leg (distinct host images through submit() / result(), outside `value`); 0 skips it")
    ap.add_argument("--alt-issue", default=os.environ.get("DD3D_BENCH_ALT_ISSUE", "2x10"),
                    help="a second issue geometry SLOTSxMICROBATCH timed in the same run and reported as config.alt_issue ('' skips it)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r06_tower_hbm_bytes.json"),
                    help="PMC-derived HBM bytes per launch of the dominant kernel, keyed by kernel signature (see profiles/README.md)")
    return ap.parse_args()


def kernel_time_us(plan, op, iters=5, burst=8):
    """Mean duration of one launch of ``op``: HIP events recorded on the launch stream (= torch's current stream) around a
    burst of back-to-back launches, so the figure is the kernel's own duration (as rocprofv3 reports it), not the
    host-side launch gap of a lone eager launch."""
    from dd3d_amd import hip
    st = hip.current_stream()
    for _ in range(3):
        op(plan.lib, st)
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(burst):
            op(plan.lib, st)
        e1.record()
        e1.synchronize()
        t += e0.elapsed_time(e1)
    return t / (iters * burst) * 1e3


def mfma_ceiling_tflops(blocks=256, iters=1500):
    """The matrix pipe's rate on realistic operand bits, measured NOW on this chip (tests/tools/src/mfma_probe.hip: the tower kernel's MFMA
    instruction and wave tile on register-resident operands, no memory traffic): the two-half-term planes of gaussian values, half of the
    activation (A) values zero as after a ReLU.  The data decides the rate (zeros: 0.98 of nominal), the chip and its thermal state the rest."""
    import ctypes as C
    from dd3d_amd import hip
    # (bench / test tooling, not the product library: tests/tools/src/mfma_probe.hip -> tests/tools/lib/libdd3d_tools.so, built by
    # __graft_entry__.build(); absent -> the caller reports no measured ceiling)
    lib = C.CDLL(os.path.join(ROOT, "tests", "tools", "lib", "libdd3d_tools.so"))
    lib.dd3d_tools_mfma_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((blocks * 512, 4, 4, 8), device="cuda", generator=g) * 4.0  # [thread][set][A row 0, A row 1, B col 0, B col 1][8]
    x[:, :, :2] *= (torch.rand(x[:, :, :2].shape, device="cuda", generator=g) < 0.5)
    hi = x.half()
    lo = (x - hi.float()).half()
    ops = torch.stack([hi[:, :, 0], hi[:, :, 1], lo[:, :, 0], lo[:, :, 1], hi[:, :, 2], hi[:, :, 3], lo[:, :, 2], lo[:, :, 3]], 2).contiguous()
    sink = torch.zeros(1, device="cuda")
    st = hip.current_stream()
    best = None
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if lib.dd3d_tools_mfma_probe(ops.data_ptr(), blocks, iters, sink.data_ptr(), st) != 0:
            raise RuntimeError("dd3d_tools_mfma_probe failed")
        e1.record()
        e1.synchronize()
        if rep:  # the first launch ramps the clocks
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
    return 2.0 * 32 * 32 * 16 * 48 * iters * blocks * 8 / (best * 1e-3) / 1e12


def main():
    args = parse_args()
    from dd3d_amd import build_model, get_cfg, hip
    from dd3d_amd.engine import ConvOp
    from dd3d_amd.parallel import DistributedForward, PipelinedForward, init_distributed
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    import torch.distributed as dist

    rank, local, world = init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    dev = torch.device("cuda", local)

    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_kitti"))
    model.load_state_dict(sd)
    model.math = args.math
    B = args.batch
    # Request j of a slot's micro-batch is its own synthetic image (seed 1000 + ...): the four images a timed launch plan convolves are
    # DIFFERENT images, and the parity report below compares them -- the timed plan's own detections -- with the oracle (round-5 verdict:
    # the parity of the line was that of the one-image plan, not of the plan `value` is measured on)
    request_inputs = [make_inputs(B, args.height, args.width, seed=_request_seed(rank, j, B, args.microbatch)) for j in range(max(1, args.microbatch))]
    inputs = request_inputs[0]
    pipeline_error = None
    if args.pipeline > 0:
        try:
            runner = PipelinedForward(model, B, *_padded(model, args.height, args.width), depth=args.pipeline,
                                      compute_streams=min(args.compute_streams, args.pipeline), microbatch=args.microbatch)
            plan = runner.plan
            runner.stage_all(request_inputs)
        except Exception as e:  # symmetric across ranks (same code, same sizes): every rank falls back together; reported in the JSON
            pipeline_error = f"{type(e).__name__}: {e}"
            args.pipeline = 0
    if args.pipeline <= 0:
        runner = DistributedForward(model, B, *_padded(model, args.height, args.width), use_graph=not args.no_graph)
        plan = runner.plan
        model.stage_inputs(inputs, plan=plan)  # H2D once: inputs are resident in HBM when the timed region starts
    torch.cuda.synchronize()

    backend = dist.get_backend() if world > 1 else None
    from dd3d_amd.parallel import ensure_exchange_ready
    transport = ensure_exchange_ready() if world > 1 else None  # (the runner ran it before capturing its graphs; cached)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")  # (gloo: the host-staged test transport)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    flush = getattr(runner, "flush", lambda: None)  # (a partly filled micro-batch slot runs in full before the clock stops)
    # untimed warm-up: --warmup steps, but never fewer than two rounds over every slot of the pipeline (round 2's driver command,
    # --warmup 5 on 16 slots, left 11 slots cold inside the timed block)
    # (one step at a time: at least 40 -- twice in five runs of this round the HIP runtime stalled ~50 ms once, somewhere in the first 45
    # graph replays of a fresh process (profiles/r04z_bench_serial.json, block 1); the untimed warm-up now covers that)
    warm = max(args.warmup, 2 * args.pipeline * args.microbatch) if args.pipeline > 0 else max(args.warmup, 40)
    for _ in range(warm):
        runner.step()
    flush()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    flush()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_local)
    ms_per_step = elapsed / args.steps * 1e3
    # N > 1: every rank's own clock over the timed block and the device it ran on -- lets a reader of the line confirm that RCCL saw N
    # DISTINCT GPUs and that no rank idled (value itself uses the max over ranks, as the contract says)
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "ms_per_step": round(elapsed_local / args.steps * 1e3, 4)})
    value = world * B * args.steps / elapsed

    # the same timed block repeated (not `value`): spread of box / clock state within one run
    block_ms = [ms_per_step]
    for _ in range(max(0, args.repeat_blocks)):
        barrier()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(args.steps):
            runner.step()
        flush()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        block_ms.append(max_over_ranks(time.perf_counter() - tb) / args.steps * 1e3)

    # for transparency: the same forward issued strictly one slot at a time on one stream (not part of the timed region)
    serial_ms = None
    if world == 1 and args.pipeline > 0:
        slot = runner.slots[0]
        for _ in range(5):
            slot.pre_graph.replay()
            slot.post_graph.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            slot.pre_graph.replay()
            slot.post_graph.replay()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / 50 * 1e3 / args.microbatch  # (one replay covers `microbatch` requests)

    # The TRUE bs=1 path (BASELINE.json configs[1] read literally: one image per launch, one request at a time, nothing else in flight):
    # its own one-image launch plan, one hipGraph replay per image, the host waiting for each.  Not part of the timed region.
    bs1_ms = None
    if world == 1 and args.pipeline > 0 and not args.no_graph:
        one = model.get_plan(B, *_padded(model, args.height, args.width))
        model.stage_inputs(inputs, plan=one)
        if one.graph is None:
            one.capture()
        for _ in range(30):
            one.run()
        torch.cuda.synchronize()
        reps = 40
        t1 = time.perf_counter()
        for _ in range(reps):
            one.run()
            torch.cuda.synchronize()  # one request at a time: the next image is not issued before this one's detections exist
        bs1_ms = (time.perf_counter() - t1) / reps * 1e3
        one.check_status()

    for pl in ([sl.plan for sl in runner.slots] if hasattr(runner, "slots") else [plan]):
        pl.check_status()  # a half-range overflow of the f16x2 arithmetic would invalidate the run: fail loudly
    work_verified = verify_work(runner, plan, B, max(1, args.microbatch) if args.pipeline > 0 else 1)
    # ---- outside `value`: the reference forward's own boundary (core.py:65 `.to(device)` ... core.py:153-164 Instances), end to end
    e2e = None
    if world == 1 and args.pipeline > 0 and args.e2e_requests > 0:
        e2e = {src: e2e_leg(model, runner, args, pinned=(src == "pinned")) for src in ("pinned", "pageable")}
        runner.stage_all(request_inputs)  # (the slots hold the e2e leg's images now: put the bench requests back for the parity report)
        runner.synchronize()
        for _ in range(args.pipeline * max(1, args.microbatch)):
            runner.step()
        runner.synchronize()
    alt_issue = None
    if world == 1 and args.pipeline > 0 and args.alt_issue:
        alt_issue = alt_issue_leg(model, args, B)
    from dd3d_amd.engine import MATH_NAMES, kernel_signature
    math_name = next(k for k, v in MATH_NAMES.items() if v == plan.math)
    peak = PEAK_F32_MFMA_TFLOPS if math_name == "f32" else PEAK_BF16_MFMA_TFLOPS / PRODUCTS[math_name]
    srt = sorted(block_ms)
    out = {
        "metric": "images/sec (384x1280) DD3D-DLA34 fwd", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPES[math_name], "data": "synthetic",
        "config": {
            "workload": f"DD3D-DLA34 KITTI3D {args.height}x{args.width} bs={B}/GPU fp32 inference (BASELINE.json configs[1]); "
                        "uint8 image in HBM -> normalise/pad -> DLA-34 -> FPN P3-P7 -> FCOS2D/3D heads -> select/decode -> NMS",
            "global_batch": world * B, "parallelism": f"dp{world}" + ("+rccl_allgather_candidates" if world > 1 else ""),
            "hip_graph": not args.no_graph, "pipeline_slots": args.pipeline, "compute_streams": args.compute_streams if args.pipeline else 1,
            "microbatch": args.microbatch if args.pipeline else 1, "warmup_steps_run": warm, "gflop_per_image": GFLOP_PER_IMAGE,
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "achieved_tflops_per_gpu": round(value / world * GFLOP_PER_IMAGE / 1e3, 2),
            "math": math_name, "split_planes": bool(plan.use_planes),
            "issue": (f"{args.pipeline} plan slots on {min(args.compute_streams, args.pipeline)} compute streams + 1 exchange/NMS stream "
                      f"(dd3d_amd.parallel.PipelinedForward); a slot's launch plan covers {args.microbatch} queued single-image request(s) "
                      "(a step = one request of bs images; the slot is enqueued when its last request arrives, a partly filled slot is "
                      "flushed -- and runs in full -- before the clock stops); every step does all of its work and all K steps are "
                      "complete at the closing synchronize") if args.pipeline else "one step at a time",
            "pipeline_error": pipeline_error,
            # one slot's two graph halves replayed back to back on one stream, nothing else in flight (per request: / microbatch)
            "ms_per_step_one_slot_at_a_time": None if serial_ms is None else round(serial_ms, 4),
            "images_per_s_one_slot_at_a_time": None if serial_ms is None else round(B / serial_ms * 1e3, 2),
            "frac_of_mfma_peak_whole_forward": round(value / world * GFLOP_PER_IMAGE / 1e3 / peak, 4),
            # configs[1] read literally -- ONE image per launch, one request at a time, the host waiting for each (its own bs-image plan):
            "bs1_ms_per_image": None if bs1_ms is None else round(bs1_ms / B, 4),
            "bs1_images_per_s": None if bs1_ms is None else round(B / bs1_ms * 1e3, 2),
            # what a request waits in the issue mode `value` is measured in: from the moment its slot is full, the slot's forward alone
            # on the chip; with every slot in flight (the timed region), slots x microbatch requests are in the system (Little's law)
            "request_latency_ms": None if (serial_ms is None or not args.pipeline) else {
                "slot_forward_alone": round(serial_ms * args.microbatch, 4),
                "pipeline_full": round(args.pipeline * args.microbatch * ms_per_step, 4),
                "one_image_at_a_time": None if bs1_ms is None else round(bs1_ms, 4),
                "note": "a request of the shipped mode also waits for its slot's micro-batch to fill (arrival-rate dependent, not included)"},
            "transport": None if world == 1 else ("rccl (torch.distributed backend nccl)" if backend == "nccl" else
                                                  f"{backend}: host-staged TEST transport of the N > 1 code path -- NOT RCCL, not a scaling number"),
            # N > 1: the start-up self-test of the exchange (dd3d_amd.parallel.exchange_selftest: one stamped all_gather + checksum, before any
            # graph capture) and what it found: ranks the transport carried, every rank's device, how many DISTINCT GPUs
            "rccl_nranks": None if world == 1 else (transport["nranks"] if backend == "nccl" else 0),
            "exchange_selftest": None if world == 1 else {k: transport[k] for k in ("nranks", "backend", "ms", "distinct_devices")},
            "rank_devices": None if world == 1 else [{k: d.get(k) for k in ("host", "device", "pci_bus_id", "name", "visible")} for d in transport["devices"]],
            "rank_ms_per_step": None if per_rank is None else [r["ms_per_step"] for r in sorted(per_rank, key=lambda r: r["rank"])],
            "graph_exchange": None if world == 1 else bool(getattr(runner, "step_graph", None) is not None),
        },
        "work_verified": work_verified,
        "blocks": {"n": len(block_ms), "steps_each": args.steps, "ms_per_step": [round(x, 4) for x in block_ms],
                   "median_ms_per_step": round(srt[len(srt) // 2], 4), "min_ms_per_step": round(srt[0], 4), "max_ms_per_step": round(srt[-1], 4),
                   "median_images_per_s": round(world * B / srt[len(srt) // 2] * 1e3, 2),
                   "block0_over_median": round(block_ms[0] / srt[len(srt) // 2], 4),
                   "note": "block 0 is the timed region `value` comes from; the others repeat it"},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: the head-tower launches (4 per forward, 66 % of all FLOPs)
        towers = [op for op in plan.ops if isinstance(op, ConvOp) and op.name.startswith("towers.")]
        us = sum(kernel_time_us(plan, op) for op in towers) / len(towers)
        flops = 2.0 * towers[0].macs  # algorithmic: 2 * (sum over levels of B*H*W) * 3 towers * 256 * (9*256)
        achieved = flops / (us * 1e-6) / 1e12
        kname = kernel_signature(towers[0])
        # HBM bytes of one launch from the PMC passes of this kernel (profiles/README.md); only when the record is for THIS instantiation
        traffic, traffic_src = None, None
        if os.path.exists(args.traffic_json):
            try:
                # (a record is for one launch geometry: the same instantiation on another number of images per launch moves other bytes)
                rec = json.load(open(args.traffic_json)).get(f"{kname} @ {plan.B} images per launch")
                if rec is not None:
                    traffic, traffic_src = rec["hbm_bytes_per_launch"], f"{os.path.relpath(args.traffic_json, ROOT)} ({rec.get('collected', '')})"
            except Exception:
                traffic = None
        np_ = hip.MATH_PLANES[plan.math]
        m_rows = towers[0].info["M"]
        layers = len(getattr(towers[0], "parts", None) or [0])  # (a chain launch covers all four tower layers: dd3d_conv_launch.chain)
        m_rows = m_rows // layers
        alg_bytes = (m_rows * 256 * 2 * np_ * 2 + 3 * 2304 * 256 * 2 * np_) * layers if plan.use_planes else None  # per layer: planes in + planes out + split filters
        # What the matrix pipe sustains on REAL operand bits (dd3d_tools_mfma_probe, timed here; stand-alone: tests/tools/src/mfma_power_bench.hip,
        # profiles/r03_mfma_power_bench.txt, r03l_*): the same v_mfma_f32_32x32x16_f16 stream from registers, no memory traffic, reaches
        # 2.45 PFLOP/s on zeros / constants and 1.45-1.77 PFLOP/s (chip and thermal state) on the two-half-term planes of gaussian data
        # with half of the activations zero -- the chip's power management, not the kernel.
        try:
            measured_mfma_ceiling_tflops = round(mfma_ceiling_tflops(), 1)
        except Exception:  # (a library without the probe: nothing was measured in this run, and nothing is reported as if it had been)
            measured_mfma_ceiling_tflops = None
        out["roofline"] = {
            "kernel": kname + (f" (head towers: {layers} layers x 15 segments in ONE chain launch)" if layers > 1 else " (head towers, 15 segments / launch)"), "bound": "mfma",
            "tower_layers_per_launch": layers,
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_basis": ("157.3 TFLOP/s dense f32-input MFMA" if math_name == "f32" else
                           f"2500 TFLOP/s dense 16-bit MFMA / {PRODUCTS[math_name]} matrix products per f32 product ({math_name}); f32-equivalent FLOP/s. "
                           "For reference the f32-input MFMA peak is 157.3"),
            "executed_16bit_mfma_tflops": None if math_name == "f32" else round(achieved * PRODUCTS[math_name], 1),
            "measured_mfma_ceiling_on_real_operands_tflops": None if math_name == "f32" else measured_mfma_ceiling_tflops,
            "frac_of_measured_ceiling": (None if math_name == "f32" or not measured_mfma_ceiling_tflops else
                                         round(achieved * PRODUCTS[math_name] / measured_mfma_ceiling_tflops, 4)),
            "ceiling_note": "`frac` is against the nominal dense peak as the contract asks; a register-resident loop of the same MFMA instruction "
                            "(dd3d_tools_mfma_probe, timed in this run) reaches 0.98 of that peak on zero operands and 0.58-0.71 on realistic ones, "
                            "depending on the chip and its thermal state (profiles/r03_mfma_power_bench.txt, r03l_mfma_power_bench_orders.txt)",
            "images_per_launch": plan.B,
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
            "flops_per_launch": flops, "avg_launch_us": round(us, 2), "tile": list(towers[0].info["tile"]),
            "blocks": towers[0].info["blocks"],
        }
        out["config"]["e2e"] = e2e
        out["config"]["alt_issue"] = alt_issue
        out["config"]["f16x2_range"] = _headroom([sl.plan for sl in runner.slots] if hasattr(runner, "slots") else [plan])
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], oracle_outs = cpu_baseline(cfg, sd, args, B)
            # the metric's second half ("3D-box L1 vs ref"): the detections of the TIMED launch plan (slot 0, every position the CPU leg
            # has an oracle forward for) against the oracle on the same uint8 images (tests/parity.py; outside the clock)
            out["parity"] = parity(model, cfg, runner, plan, request_inputs, oracle_outs, B, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _request_seed(rank, j, B, microbatch):
    """Seed of request j (position j of every slot) on `rank`: B consecutive seeds per request, disjoint over positions and ranks."""
    return 1000 + (rank * max(1, microbatch) + j) * B


def _padded(model, H, W):
    d = model.backbone.size_divisibility
    return (H + d - 1) // d * d, (W + d - 1) // d * d


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box exposes 256 logical
    CPUs to a container that is only allowed a fraction of them; 256 torch threads on that quota thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, min(n, 32))


def verify_work(runner, plan, B, microbatch=1):
    """After the timed blocks (round-4 verdict: nothing was read back from the timed work): request j -- its own image -- was staged into
    position j of EVERY slot, so position j of every slot must hold the SAME detections as position j of slot 0 -- counts and every field,
    bit for bit (same kernels, same tile choices) -- and they must exist.  A slot whose work was skipped, or ran on stale inputs, fails
    here.  (What slot 0 holds is compared with the oracle by `parity`.)"""
    plans = [sl.plan for sl in runner.slots] if hasattr(runner, "slots") else [plan]
    ref, checked = {}, 0
    for p in plans:
        counts = p.det_count.cpu()
        for g in range(p.det.shape[0]):
            key = g % (B * microbatch)
            n = int(counts[g])
            d = p.det[g, :n].cpu()
            if key not in ref:
                ref[key] = (n, d)
            elif n != ref[key][0] or not torch.equal(d, ref[key][1]):
                raise RuntimeError(f"bench: image {g} of a slot holds other detections than slot 0 ({n} vs {ref[key][0]}): a step did not do its work")
            checked += 1
    if not ref:
        return {"slot_positions_checked": 0, "detections_per_image": [], "identical_across_slots": True, "note": "this plan owns no images"}
    if not all(n > 0 for n, _ in ref.values()):
        raise RuntimeError("bench: a synthetic image produced no detections: decode / NMS were not exercised")
    return {"slot_positions_checked": checked, "detections_per_image": [ref[k][0] for k in sorted(ref)], "distinct_images_per_slot": len(ref),
            "identical_across_slots": True}


def _headroom(plans):
    """Worst f16x2 range headroom over the plans that ran (dd3d_amd.engine.PlanBase.range_headroom): how near the per-element overflow
    guard (and the bf16x3 fallback behind it) the run came."""
    rows = [h for h in (p.range_headroom() for p in plans) if h]
    if not rows:
        return None
    worst = min(rows, key=lambda h: h["overflow_headroom_x"])
    low = min(rows, key=lambda h: h["underflow_headroom_x"])
    r = lambda x: round(float(x), 4)
    return {"plane_scale": worst["plane_scale"], "largest_activation": r(worst["largest_activation"]), "overflow_at": r(worst["overflow_at"]),
            "overflow_headroom_x": r(worst["overflow_headroom_x"]), "largest_in": worst["largest_in"],
            "underflow_headroom_x": r(low["underflow_headroom_x"]), "smallest_in": low["smallest_in"], "launches_watched": worst["launches_watched"],
            "note": "sampled per-launch maxima of |activation| (a lower bound: one wave tile per block); the guard trips per element at overflow_at"}


def parity(model, cfg, runner, plan, request_inputs, oracle_outs, B, args):
    """tests/parity.py::parity_report of the TIMED launch plan: slot 0 of the pipeline (or the one plan of `--pipeline 0`) after the timed
    blocks, image 0 of every request position the CPU leg produced an oracle forward for, against that forward.  Belongs to the
    cpu_baseline leg: the oracle is the checker here, never the thing measured.  An installed dd3d_amd without the tests/ tree reports
    `parity: null` instead of failing after all the timing is done (round-5 advisor)."""
    try:
        from tests.parity import parity_report
    except ImportError as e:
        return {"pass": None, "note": f"tests/parity.py is not importable here ({e}): no parity report"}
    p = runner.slots[0].plan if hasattr(runner, "slots") else plan
    torch.cuda.synchronize()
    reps = []
    for j, (ref, stages) in sorted(oracle_outs.items()):
        if j >= len(request_inputs) or (j + 1) * B > p.B:
            continue
        inp = request_inputs[j]
        sizes = [(int(x["image"].shape[-2]), int(x["image"].shape[-1])) for x in inp]
        out = model.collect(p, inp, sizes, first=j * B)
        reps.append(parity_report(out[0], ref[0], plan=p, stages=stages, cfg=cfg, image=j * B, ref_image=0))
    if not reps:
        return {"pass": None, "note": "the CPU leg ran no oracle forward"}
    worst = lambda k: max(r[k] for r in reps if k in r) if any(k in r for r in reps) else None
    total = lambda k: sum(r.get(k, 0) for r in reps)
    rep = {"images_compared": len(reps), "detections_hip": total("detections_hip"), "detections_oracle": total("detections_oracle"),
           "matched": total("matched"), "int_mismatches": total("int_mismatches"), "candidates_hip": total("candidates_hip"),
           "candidates_oracle": total("candidates_oracle"), "on_cut_flips": total("on_cut_flips"), "off_cut_flips": total("off_cut_flips"),
           "max_flip_margin": worst("max_flip_margin"), "rank_swaps": total("rank_swaps"), "rank_swap_gap_rel_max": worst("rank_swap_gap_rel_max")}
    for k in ("box3d_l1_tvec_size", "box3d_l1_tvec_size_rel", "corners_l1", "corners_l1_rel", "depth_rel_max", "size_rel_max", "box2d_abs_max",
              "box2d_rel_max", "score_rel_max", "score3d_rel_max", "quat_abs_max_up_to_sign"):
        rep[k] = worst(k)  # (the worst image's figure)
    rep = {k: (round(v, 9) if isinstance(v, float) else v) for k, v in rep.items()}
    rep["tolerance_rel"] = reps[0]["tolerance_rel"]
    rep["pass"] = all(r["pass"] for r in reps)
    rep["per_image_pass"] = [bool(r["pass"]) for r in reps]
    rep["image"] = (f"image 0 of request positions {sorted(oracle_outs)[:len(reps)]} of slot 0's launch plan after the timed blocks: the {p.B}-image plan `value` "
                    f"is measured on (tile table entries of {p.B} images per launch), distinct synthetic images (seeds {[_request_seed(0, j, B, args.microbatch) for j in sorted(oracle_outs)[:len(reps)]]}), "
                    "default arithmetic of the run; floats: the worst image's")
    rep["reference"] = "oracle/dd3d_oracle.py forward of the same uint8 images (cpu_baseline leg); quantities: boxes3d.py:47-64 corners, :142-144 vectorize"
    return rep


def e2e_leg(model, runner, args, pinned=True):
    """What the reference's forward does, inside the clock (tridet/modeling/dd3d/core.py:65 `x["image"].to(self.device)` ... :153-164 the
    returned Instances): every request is a DISTINCT host image handed to `runner.submit()`, its result comes back through
    `runner.result()` as Instances -- H2D copy, `stage_inputs`, the slot's graphs, `collect`, all of it.  Results are collected with a lag
    of (slots - 1) x microbatch requests, as a data-loader-fed evaluation loop would.  NOT `value` (the contract times inputs resident in
    HBM); reported beside it."""
    from collections import deque
    from dd3d_amd.synthetic import make_inputs
    B, mb, depth = args.batch, max(1, args.microbatch), args.pipeline
    n_pool = depth * mb + mb  # more distinct images than requests in flight
    pool = [make_inputs(B, args.height, args.width, seed=50000 + i * B) for i in range(n_pool)]
    if pinned:
        for req in pool:
            for x in req:
                x["image"] = x["image"].pin_memory()
    lag = max(1, (depth - 1) * mb)  # (the largest lag the runner allows: a slot's results must be taken before the slot is filled again)
    host = {"stage": 0.0, "collect": 0.0, "submit": 0.0, "result": 0.0}
    stage_inputs, collect = model.stage_inputs, model.collect

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            host[name] += time.perf_counter() - t
            return r
        return w

    model.stage_inputs, model.collect = timed("stage", stage_inputs), timed("collect", collect)
    try:
        def run(n):
            q, ndet = deque(), 0
            submit, result = timed("submit", runner.submit), timed("result", runner.result)
            for i in range(n):
                q.append(submit(pool[i % n_pool]))
                if len(q) > lag:
                    ndet += sum(len(o["instances"]) for o in result(q.popleft()))
            while q:
                ndet += sum(len(o["instances"]) for o in result(q.popleft()))
            return ndet

        run(2 * depth * mb)  # warm-up: every slot twice
        torch.cuda.synchronize()
        for k in host:
            host[k] = 0.0
        n = max(args.e2e_requests // mb, 1) * mb
        t0 = time.perf_counter()
        ndet = run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        model.stage_inputs, model.collect = stage_inputs, collect
    return {"images_per_s": round(n * B / dt, 2), "ms_per_request": round(dt / n * 1e3, 4), "requests": n, "detections_returned": ndet,
            "host_us_per_request_stage_inputs": round(host["stage"] / n * 1e6, 1), "host_us_per_request_collect": round(host["collect"] / n * 1e6, 1),
            # submit() = stage_inputs + slot bookkeeping + (every microbatch-th request) the slot's graph launches; result() = waiting for the
            # slot's post half + collect: what is left of a request's wall time after these two is the loop itself
            "host_us_per_request_submit": round(host["submit"] / n * 1e6, 1), "host_us_per_request_result": round(host["result"] / n * 1e6, 1),
            "host_us_per_request_waiting_for_gpu": round((host["result"] - host["collect"]) / n * 1e6, 1),
            "source": "pinned host memory" if pinned else "pageable host memory", "collect_lag_requests": lag,
            "covers": "H2D of a distinct uint8 image per request + stage_inputs + the slot's hipGraphs + collect -> Instances (core.py:65 ... :153-164)"}


def alt_issue_leg(model, args, B):
    """The same K steps issued in another geometry (default 2 slots x 10 requests per launch plan: profiles/r05s_coalescing_sweep.txt),
    timed in the same run on the same chip state -- reported, not `value`."""
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import make_inputs
    try:
        depth, mb = (int(v) for v in args.alt_issue.lower().split("x"))
        r = PipelinedForward(model, B, *_padded(model, args.height, args.width), depth=depth, compute_streams=min(args.compute_streams, depth), microbatch=mb)
        r.stage_all([make_inputs(B, args.height, args.width, seed=_request_seed(0, j % max(1, args.microbatch), B, args.microbatch)) for j in range(mb)])
        for _ in range(2 * depth * mb):
            r.step()
        r.flush()
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                r.step()
            r.flush()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / args.steps * 1e3)
        for sl in r.slots:
            sl.plan.check_status()
        ms.sort()
        return {"issue": f"{depth} slots x {mb} requests per launch plan", "steps": args.steps, "ms_per_step_blocks": [round(x, 4) for x in ms],
                "median_images_per_s": round(B / ms[1] * 1e3, 2),
                "note": f"{args.steps} steps = {-(-args.steps // mb)} slot run(s) of {mb} (a partly filled last slot runs in full); not `value`"}
    except Exception as e:
        return {"issue": args.alt_issue, "error": f"{type(e).__name__}: {e}"}


def cpu_baseline(cfg, sd, args, B=1, budget_s=12.0):
    """The oracle (oracle/dd3d_oracle.py, a torch-CPU fp32 restatement of the reference forward) on the host cores:
    one warm-up at 1/16 of the pixels, then single-image forwards of the bench workload -- image 0 of request 0, 1, ... of the timed
    micro-batch, cycling -- until ~``budget_s`` of CPU time is spent (at least one, at most ``--cpu-forwards``).  Returns the baseline
    record and {request position: (oracle result, oracle stages)} for the parity report."""
    from dd3d_amd.synthetic import make_inputs
    from oracle import dd3d_oracle as O
    threads = usable_cores()
    torch.set_num_threads(threads)
    mb = max(1, args.microbatch) if args.pipeline > 0 else 1
    images = [make_inputs(1, args.height, args.width, seed=_request_seed(0, j, B, args.microbatch)) for j in range(mb)]
    outs = {}
    with torch.no_grad():
        O.dd3d_forward(sd, cfg, make_inputs(1, max(128, args.height // 4 // 128 * 128), max(128, args.width // 4 // 128 * 128)))
        n, t0 = 0, time.perf_counter()
        while n < args.cpu_forwards and (n == 0 or time.perf_counter() - t0 < budget_s * n / (n + 1)):
            j = n % mb
            last = O.dd3d_forward(sd, cfg, images[j])
            if j not in outs:
                outs[j] = last
            n += 1
        dt = (time.perf_counter() - t0) / n
    return {
        "value": round(1.0 / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{n} forward(s) of 1 synthetic {args.height}x{args.width} image each ({len(outs)} distinct image(s) of the timed micro-batch, after a "
                  f"small warm-up), oracle/dd3d_oracle.py on torch {torch.__version__} CPU fp32 with {threads} threads (os.cpu_count()={os.cpu_count()})",
    }, outs


if __name__ == "__main__":
    main()
