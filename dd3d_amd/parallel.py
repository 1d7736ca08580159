"""Multi-GPU forward: one process per MI355X, images sharded across ranks, ONE collective per step.

The reference's inference uses no device collective (each rank evaluates its shard, pickled results are gathered
over gloo at evaluation time: tridet/data/build.py:75-93, kitti_3d_evaluator.py:154-161).  The north star adds one:
every rank decodes the candidates of its own images, then ONE RCCL ``all_gather_into_tensor`` (xGMI) of each rank's
record -- [candidates | per-level counts | resize targets], fixed capacity, <= 330 KB per KITTI image -- precedes the
batched NMS, which every rank then runs for the images it owns, out of its segment of the gathered buffer (no rank
repeats another rank's NMS).  The payload is latency-bound: one call, no bucketing.

Rank r decodes the global images ``[r*B, (r+1)*B)`` (rank-major order == gather order).  Who READS the gathered records:

* ``camera_sharded=True`` (NuscenesDD3D; the north star's "images shard one-per-GPU"): the cameras of a sample sit on different ranks --
  global images 6 s .. 6 s + 5 are sample s -- and the rank that decoded a sample's first camera OWNS it: it runs the 2D NMS of all six
  cameras and the sample-level BEV aggregation (nuscenes_dd3d.py:448-465, postprocessing.py:58-108) on the records the other ranks
  delivered (the record carries K^-1 and the camera->global pose of every image beside its candidates), and returns the sample's
  detections.  Here the exchange is REQUIRED: without it no rank holds all cameras of a sample.
* otherwise a rank finalises its own B images out of ITS segment of the gathered buffer (the samples are rank-local, as the
  reference's InferenceGroupSampler keeps them, group_sampler.py:30-35).  Nothing then consumes the other ranks' records: the
  collective is what the north star prescribes for the step ("an RCCL gather of decoded boxes before batched NMS") and what bench.py
  measures, but a deployment that only needs per-image results can drop it (``exchange=False``: one graph, no collective).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend is None:
            # DD3D_DIST_BACKEND=gloo: test transport for driving the N > 1 path with several ranks on ONE GPU (host-staged gather);
            # the ranks then share the visible devices round-robin
            backend = os.environ.get("DD3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if backend != "nccl":
                local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def owner_of_image(g, B):
    """Rank that produced global image g."""
    return g // B


def inference_shard(total_size, group_size, rank, world_size):
    """Dataset indices rank `rank` evaluates -- the arithmetic of the reference's InferenceGroupSampler
    (tridet/data/samplers/group_sampler.py:14-35): the dataset is a sequence of in-order groups (group_size = 1 for KITTI, the 6
    cameras of a sample for nuScenes); every rank takes a contiguous run of WHOLE groups, ceil(groups / world) of them, the last ranks
    possibly fewer or none.  A rank's per-step batch must then be a multiple of group_size, which is what keeps a sample's cameras
    on one rank for the BEV aggregation that follows the gather."""
    assert total_size > 0 and group_size > 0
    assert total_size % group_size == 0, f"The total size must be divisible by group size: total size={total_size}, group size={group_size}"
    num_groups = total_size // group_size
    shard_size = ((num_groups - 1) // world_size + 1) * group_size
    begin = min(shard_size * rank, total_size)
    return range(begin, min(shard_size * (rank + 1), total_size))


def padded_inference_shard(total_size, group_size, rank, world_size, batch_size):
    """`inference_shard` for runners that hold a collective in every step: (indices, valid) of equal length ON EVERY RANK.

    The reference's sampler hands the tail ranks fewer items or none (KITTI val on 8 GPUs: 7 x 472 + 465), and its forward has no
    collective, so ranks may stop early.  Here every step of every rank takes part in one all_gather: a rank that stops early -- or the
    short final batch of a shard -- would leave the others waiting in RCCL until the watchdog fires.  This helper pads every rank's
    shard to the SAME number of whole batches, steps = ceil(shard_size / batch_size) of the largest shard; padding entries carry
    valid = False, the caller feeds them like any other image and drops their outputs (`DistributedForward.forward(batch, valid=...)`
    does).  batch_size must be a multiple of group_size so that a nuScenes sample never straddles two steps.

    **No batch ever holds a group twice** (round-3 advisor: repeating the rank's last group put one nuScenes sample twice into a padded
    batch of two or more groups; NuscenesDD3D's sample grouping -- nuscenes_dd3d.py:77-87 get_group_idxs -- then sees 12 images under one
    sample_token and raises on the tail rank only, while its peers wait in the step's collective).  Padding groups are whole dataset groups
    that the batch does not hold yet: first the rank's own, from the start of its shard, then -- a shard with fewer groups than a batch --
    the dataset's, in order."""
    assert batch_size > 0 and batch_size % group_size == 0, "a step's batch must hold whole groups"
    own = list(inference_shard(total_size, group_size, rank, world_size))
    largest = len(inference_shard(total_size, group_size, 0, world_size))  # rank 0 always holds a full shard
    steps = -(-largest // batch_size)
    per_batch, num_groups = batch_size // group_size, total_size // group_size
    if per_batch > num_groups:
        raise ValueError(f"a step's batch holds {per_batch} groups but the dataset has only {num_groups}: no batch can be filled with distinct "
                         f"groups -- use batch_size <= {num_groups * group_size}")
    own_groups = [i // group_size for i in own[::group_size]]
    own_set = set(own_groups)
    pool = own_groups + [g for g in range(num_groups) if g not in own_set]  # candidates for padding, in order of preference
    idx, valid = [], []
    for st in range(steps):
        real = own_groups[st * per_batch:(st + 1) * per_batch]
        fill = []
        if len(real) < per_batch:  # (only the tail of a shard: a full batch never walks the pool)
            real_set = set(real)
            for g in pool:
                if g not in real_set:
                    fill.append(g)
                    if len(real) + len(fill) == per_batch:
                        break
        for g, v in [(g, True) for g in real] + [(g, False) for g in fill]:
            idx += list(range(g * group_size, (g + 1) * group_size))
            valid += [v] * group_size
    return idx, valid


def graph_exchange_mode():
    """DD3D_GRAPH_EXCHANGE: "0" (default) the collective runs eagerly between the step's two captured graph halves -- the form every
    torch.distributed user runs; "1" capture it INSIDE the step's hipGraph; "probe" let `graph_exchange_probe` decide at start-up.
    The default stays "0" although the one-graph step measured 1.7 % faster on one rank (profiles/r04y_*): no box with more than one GPU was
    ever available to validate a captured multi-rank RCCL collective, a probe cannot be guaranteed not to hang on a transport that does not
    support capture, and a hang in the first real N > 1 run costs more than 15 us per step ever returns."""
    v = os.environ.get("DD3D_GRAPH_EXCHANGE", "0").strip().lower()
    return v if v in ("0", "1", "probe") else "0"


_PROBE_RESULT = {}


def graph_exchange_probe(group=None, timeout_s=20.0):
    """Can THIS transport replay a captured all_gather?  Captures a tiny all_gather_into_tensor (on a communicator of its own, so that a
    failure cannot poison the one the steps use) into a hipGraph, replays it twice on changing inputs, checks what arrived on every rank
    and takes the AND over the ranks (one eager all_reduce), so that every rank reaches the same decision.  A capture error or wrong
    data -> False; a replay that does not complete within `timeout_s` -> False as well (the probe's stream and communicator are then
    abandoned).  Cached per process."""
    key = id(group)
    if key in _PROBE_RESULT:
        return _PROBE_RESULT[key]
    ok = False
    if dist.is_initialized() and dist.get_backend(group) == "nccl" and torch.cuda.is_available():
        import time
        import warnings
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device("cuda", torch.cuda.current_device())

        def agree(flag):  # AND over the ranks on the step's own (working) communicator: every rank takes the same branch below
            v = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
            return bool(int(v.item()))

        # Every rank issues the SAME sequence of collectives whatever fails locally (round-5 advisor: a rank that threw before / inside
        # new_group or the first all_gather left its peers blocked in them): the probe communicator is created unconditionally, each stage
        # records its local outcome, and the ranks agree on it before the next stage -- nobody replays a graph some rank failed to capture.
        pg = dist.new_group(ranks=list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group), backend="nccl")
        side = torch.cuda.Stream()
        src = torch.zeros(64, dtype=torch.float32, device=dev)
        dst = torch.zeros(64 * world, dtype=torch.float32, device=dev)
        why, g = None, None
        try:
            with torch.cuda.stream(side):
                src.fill_(float(rank) + 0.25)
                dist.all_gather_into_tensor(dst, src, group=pg)  # eager first: creates the communicator outside the capture
                side.synchronize()
        except Exception as e:
            why = f"eager all_gather on the probe communicator: {type(e).__name__}: {e}"
        if agree(why is None):
            try:
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        dist.all_gather_into_tensor(dst, src, group=pg)
            except Exception as e:
                why = f"capture: {type(e).__name__}: {e}"
            if agree(why is None):
                good = True
                try:
                    with torch.cuda.stream(side):
                        for it in range(2):
                            src.fill_(1000.0 * (it + 1) + rank)
                            dst.zero_()
                            g.replay()
                            ev = torch.cuda.Event()
                            ev.record(side)
                            t0 = time.perf_counter()
                            while not ev.query():
                                if time.perf_counter() - t0 > timeout_s:
                                    raise TimeoutError("captured all_gather did not complete")
                                time.sleep(0.001)
                            want = torch.arange(world, dtype=torch.float32, device=dev).repeat_interleave(64) + 1000.0 * (it + 1)
                            good = good and bool(torch.equal(dst, want))
                    if not good:
                        why = "replay delivered wrong data"
                except Exception as e:  # (a timed-out replay: the probe's stream and communicator are abandoned, not destroyed)
                    why = f"replay: {type(e).__name__}: {e}"
                ok = agree(why is None)
        if why is not None:
            warnings.warn(f"dd3d_amd: the all_gather cannot be captured in a hipGraph on this transport ({why}); "
                          "the exchange runs eagerly between the two graph halves")
        if why is None or not why.startswith("replay: TimeoutError"):
            try:
                dist.destroy_process_group(pg)  # (the probe's communicator is not needed again; a hung one is left alone)
            except Exception:
                pass
    _PROBE_RESULT[key] = ok
    return ok


def graph_exchange_enabled(group=None):
    """The step's all_gather is captured inside its hipGraph: DD3D_GRAPH_EXCHANGE=1, or =probe and the probe passed; a device transport
    (RCCL) in any case."""
    mode = graph_exchange_mode()
    if mode == "0" or not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return False
    return True if mode == "1" else graph_exchange_probe(group)


def device_identity():
    """What tells two ranks' GPUs apart: host, HIP device ordinal, PCI bus id, name, memory -- gathered by `exchange_selftest` so that a
    multi-GPU run can show that RCCL saw N DISTINCT devices."""
    import socket
    d = {"host": socket.gethostname(), "pid": os.getpid(), "device": None, "pci_bus_id": None, "name": None, "total_memory_gib": None,
         "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        d.update(device=i, name=pr.name, total_memory_gib=round(pr.total_memory / 2**30, 1))
        try:  # (PyTorch >= 2.4 exposes it; the ordinal + host still identify the device without it)
            d["pci_bus_id"] = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except AttributeError:
            pass
        try:
            d["uuid"] = str(pr.uuid)
        except AttributeError:
            pass
    return d


def exchange_selftest(group=None, words=4096, timeout_s=120.0, device=None):
    """Start-up check of the step's transport, BEFORE any graph is captured: one small all_gather_into_tensor of a rank-stamped record and
    a checksum of what arrived, so that a transport that cannot carry the exchange fails fast and says why, instead of hanging the first
    step inside a hipGraph.  The wait is a polled event with a deadline (a stuck RCCL kernel would block a plain synchronize forever).
    Also gathers every rank's `device_identity`.  Returns {"nranks", "backend", "ms", "devices": [...], "distinct_devices"}; raises
    RuntimeError on a timeout, wrong data, or two ranks of an RCCL job on the same GPU.  One rank / no process group: a no-op record."""
    import time
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {"nranks": 1, "backend": None, "ms": 0.0, "devices": [device_identity()], "distinct_devices": 1}
    world, rank, backend = dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group)
    on_gpu = backend == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    stamp = lambda r: torch.arange(words, dtype=torch.float32) * 0.5 + 7919.0 * (r + 1)
    src = stamp(rank).to(dev)
    dst = torch.zeros(world * words, dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    work = dist.all_gather_into_tensor(dst, src, group=group, async_op=True)
    if on_gpu:
        ev = torch.cuda.Event()
        work.wait()  # (orders the current stream behind the collective; does not block the host)
        ev.record()
        while not ev.query():
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(
                    f"dd3d_amd exchange self-test: rank {rank}/{world} on {device_identity()} waited {timeout_s:.0f} s for one {4 * words}-byte "
                    "all_gather over RCCL.  Check that every rank sees its own GPU (LOCAL_RANK -> torch.cuda.set_device), that "
                    "HSA_ENABLE_IPC_MODE_LEGACY=0 is exported (dmabuf IPC), and that MASTER_ADDR is reachable (127.0.0.1 on one node)")
            time.sleep(0.0005)
    else:
        from datetime import timedelta
        if not work.wait(timedelta(seconds=timeout_s)):
            raise RuntimeError(f"dd3d_amd exchange self-test: rank {rank}/{world}: the {backend} all_gather did not complete in {timeout_s:.0f} s")
    ms = (time.perf_counter() - t0) * 1e3
    got = dst.cpu().view(world, words)
    bad = [r for r in range(world) if not torch.equal(got[r], stamp(r))]
    if bad:
        raise RuntimeError(f"dd3d_amd exchange self-test: rank {rank}/{world} received wrong records from rank(s) {bad} over {backend}")
    idents = [None] * world
    dist.all_gather_object(idents, device_identity(), group=group)
    keys = {(d["host"], d.get("pci_bus_id") or d["device"]) for d in idents}
    if on_gpu and len(keys) != world:
        raise RuntimeError(f"dd3d_amd exchange self-test: {world} RCCL ranks share {len(keys)} GPU(s): {idents}")
    return {"nranks": world, "backend": backend, "ms": round(ms, 3), "devices": idents, "distinct_devices": len(keys)}


_SELFTEST = {}


def ensure_exchange_ready(group=None):
    """`exchange_selftest` once per process and process group: both runners call it before they capture any graph."""
    key = id(group)
    if key not in _SELFTEST:
        _SELFTEST[key] = exchange_selftest(group)
    return _SELFTEST[key]


def gather_candidates(pairs, group=None):
    """The step's only exchange: all_gather every rank's record (ForwardPlan.gather_pairs(): one (record, [W x record]) pair) into
    the rank-major gathered buffer."""
    staged = dist.get_backend(group) == "gloo" and pairs[0][0].is_cuda
    for local, glob in pairs:
        if staged:
            # gloo has no device all_gather: test transport for driving the N > 1 GPU code path with several processes on ONE
            # GPU (tests/gpu_dist_check.py); the product transport is RCCL ("nccl"), which takes the device tensors directly
            g_cpu = torch.empty(glob.shape, dtype=glob.dtype)
            dist.all_gather_into_tensor(g_cpu, local.cpu(), group=group)
            glob.copy_(g_cpu)
        else:
            dist.all_gather_into_tensor(glob, local, group=group)


class DistributedForward:
    """Drives a ``ForwardPlan(world_size=W)``: [hipGraph: preprocess .. select/decode] -> RCCL all_gather ->
    [batched NMS of the rank's own images, read out of the gathered buffer]."""
    def __init__(self, model, B, Hp, Wp, use_graph=True, force_exchange=None, camera_sharded=False):
        self.model = model
        self.camera_sharded = bool(camera_sharded)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # one rank: the whole forward is one hipGraph, unless `force_exchange` keeps the two-phase step (RCCL check on one GPU);
        # force_exchange=False with several ranks drops the collective (per-image results only; never with camera_sharded)
        self.exchange = (self.world > 1 and force_exchange is not False) or bool(force_exchange) or self.camera_sharded
        self.B, self._geometry, self.use_graph = B, (Hp, Wp), use_graph
        self.transport = ensure_exchange_ready() if (self.exchange and self.world > 1) else None  # fails fast, before any graph is captured
        self._build()

    def _build(self):
        model, (Hp, Wp) = self.model, self._geometry
        model.use_graph = self.use_graph
        self.plan = p = model.get_plan(self.B, Hp, Wp, world_size=self.world, rank=self.rank, exchange=self.exchange, camera_sharded=self.camera_sharded)
        self.pre_graph = self.post_graph = self.step_graph = None
        if self.use_graph and self.exchange:
            # the collective sits between two captured halves
            p.launch()
            torch.cuda.synchronize()
            if graph_exchange_enabled(None):
                # DD3D_GRAPH_EXCHANGE=1 (experimental; round-3 verdict item): the RCCL all_gather is captured INSIDE the step's graph -- one
                # replay per step instead of replay / collective / replay.  The communicator is created by an eager collective first (a
                # capture must not contain its initialisation).  Validated with ONE rank only (tests/gpu_rccl_check.py graph): no
                # multi-GPU box was available to the builder, hence off by default.
                gather_candidates(p.gather_pairs())
                torch.cuda.synchronize()
                self.step_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.step_graph):
                    p.launch(0, p.num_pre_nms_ops)
                    gather_candidates(p.gather_pairs())
                    p.launch(p.num_pre_nms_ops)
                return
            self.pre_graph, self.post_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.pre_graph):
                p.launch(0, p.num_pre_nms_ops)
            with torch.cuda.graph(self.post_graph):
                p.launch(p.num_pre_nms_ops)

    def step(self):
        p = self.plan
        if not self.exchange:
            p.run()
            return
        if self.step_graph is not None:
            self.step_graph.replay()  # pre half, RCCL all_gather and post half in one captured graph
            p.fetch()
            return
        if self.pre_graph is not None:
            self.pre_graph.replay()
        else:
            p.launch(0, p.num_pre_nms_ops)
        gather_candidates(p.gather_pairs())
        if self.post_graph is not None:
            self.post_graph.replay()
        else:
            p.launch(p.num_pre_nms_ops)
        p.fetch()  # (every path that issues a forward ends with the read-back copy: what an earlier one delivered is stale from here on)

    def forward(self, batched_inputs, valid=None):
        """One step.  Every rank must call this the same number of times (each call holds the step's collective): feed the ranks with
        `padded_inference_shard`.  A batch shorter than the plan's (the last batch of a shard, or none at all) is completed with the
        images the plan's buffers already hold -- the step still runs in full -- and only the given images are returned; `valid`
        (list of bool, from padded_inference_shard) drops padding images from the result as well."""
        from dd3d_amd.engine import relax_arithmetic
        while True:
            try:
                return self._forward(batched_inputs, valid)
            except FloatingPointError as e:
                # The range guard of the default f16x2 arithmetic.  With the exchange every rank read the SAME verdict out of the gathered
                # records (engine.ForwardPlan.check_status), so all ranks arrive here on the same step and repeat it together -- first with
                # a wider half range (plane scale 16 -> 4 -> 1), then on the three-term split (engine.plan.relax_arithmetic).
                if (self.world > 1 and not self.exchange) or not relax_arithmetic(self.model, e):
                    raise
                torch.cuda.synchronize()
                self._build()

    def _forward(self, batched_inputs, valid):
        image_sizes = []
        if len(batched_inputs):
            _, image_sizes = self.model.stage_inputs(batched_inputs, plan=self.plan, partial=len(batched_inputs) < self.plan.B)
        self.step()
        if self.camera_sharded:
            # results belong to the OWNER of a sample, not to the rank that decoded a camera: [(global image index, {"instances"})]
            if self.plan.G == 0:
                torch.cuda.current_stream().synchronize()
                self.plan.check_status()  # (a rank that owns no sample still takes part in the verdict)
            return self.model.collect_owned(self.plan)
        if not len(batched_inputs):
            torch.cuda.current_stream().synchronize()
            self.plan.check_status()
            return []
        out = self.model.collect(self.plan, batched_inputs, image_sizes)
        if valid is not None:
            assert len(valid) == len(out)
            out = [o for o, v in zip(out, valid) if v]
        return out


class _DeviceRuntime:
    """What PipelinedForward needs of the device: streams, events, graph capture.  The product runtime is HIP through torch.cuda."""
    dry_run = False

    def stream(self):
        return torch.cuda.Stream()

    def event(self):
        return torch.cuda.Event()

    def on(self, stream):
        return torch.cuda.stream(stream)

    def current_stream(self):
        return torch.cuda.current_stream()

    def synchronize(self):
        torch.cuda.synchronize()

    def warm(self, plan):
        plan.launch()  # outside capture: sets kernel attributes, faults pages
        torch.cuda.synchronize()

    def capture(self, slot, plan, first, last, half):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            plan.launch(first, last)
        return g


class HostOrderRuntime(_DeviceRuntime):
    """Test runtime (tests/test_parallel_cpu.py): no device -- streams and events are no-ops, everything runs synchronously in issue order on
    dry-run launch plans, and a "graph replay" calls `pre_hook(slot)` / `post_hook(slot)` instead of kernels.  What remains is exactly the
    host logic whose ORDER matters across ranks: which slot's collective is issued when (slot ring wrap-around, micro-batches, flushes, the
    range-guard fallback's re-issue order), driven over a real process group (gloo) with several ranks."""
    dry_run = True

    class _Null:
        def wait_event(self, *a):
            pass

        wait_stream = record = synchronize = wait_event

        def query(self):
            return True

    class _Replay:
        def __init__(self, fn):
            self.replay = fn

    def __init__(self, pre_hook=None, post_hook=None):
        self.pre_hook, self.post_hook = pre_hook, post_hook

    def stream(self):
        return self._Null()

    event = current_stream = stream

    def on(self, stream):
        import contextlib
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def warm(self, plan):
        pass

    def capture(self, slot, plan, first, last, half):
        hook = self.pre_hook if half == "pre" else self.post_hook
        return self._Replay((lambda: hook(slot)) if hook is not None else (lambda: None))


class PipelinedForward:
    """Throughput mode of the same step: `depth` plan slots (each with its own activation buffers and its own pair of captured hipGraph
    halves; the packed weights are the model's, shared by all slots); the trunk + heads + select/decode of step i+1 run on a compute
    stream while the candidate exchange (RCCL) and the NMS stages of step i run on the post stream, so the collective and the
    latency-bound tail never stall the MFMA kernels.  Results are identical to `DistributedForward` (same kernels, same buffers per
    slot); only the order on the device changes.

        h = runner.submit(batched_inputs)   # stages the inputs, enqueues the slot's graphs once it is full, returns at once
        out = runner.result(h)              # waits for that request only

    `microbatch` = M > 1: a slot's launch plan covers M REQUESTS of B images each (plan batch M * B).  Requests are staged into the
    slot one after the other and the slot is enqueued when the M-th arrives (or at `flush()` / `result()` of one of its requests,
    with the unfilled positions holding whatever they held before -- their outputs are never read).  Per-request semantics are
    unchanged: every image is convolved, decoded and NMS-ed on its own, so a request's result does not depend on what shares its slot;
    what changes is that the small backbone / FPN launches cover M x the pixels each (256 CUs are not filled by one 384 x 1280
    image's coarse levels) and there are M x fewer launches per image.

    Every slot's graphs are replayed once at construction, so the first timed step of a caller does not pay a first-launch cost.

    Numeric guard (dd3d_amd.engine.PlanBase.check_status): when the default f16x2 arithmetic meets an activation outside the half pair's
    range, `result()` -- on a model on the default arithmetic -- drains the pipeline, rebuilds every slot -- with a wider half range first (plane scale 16 -> 4 -> 1), then on the three-term bf16 split --,
    re-runs the requests in flight and returns (what `DD3D.forward` does for a single forward).  With several ranks the verdict of every
    rank travels in the exchanged records, so all ranks take this path for the same slot run."""
    def __init__(self, model, B, Hp, Wp, depth=2, force_exchange=False, compute_streams=1, microbatch=1, runtime=None):
        assert depth >= 1 and compute_streams >= 1 and microbatch >= 1
        self.model = model
        self.rt = rt = runtime or _DeviceRuntime()
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.exchange = self.world > 1 or bool(force_exchange)
        self.transport = ensure_exchange_ready() if (self.exchange and self.world > 1) else None  # fails fast, before any graph is captured
        self.B, self.microbatch = B, int(microbatch)
        model._sync_flags()
        # compute_streams > 1: consecutive steps' trunks are issued on different streams and may share the chip
        self.compute_streams = [rt.stream() for _ in range(compute_streams)]
        self.post_stream = rt.stream()
        self.slots = []
        for i in range(depth):
            slot = type("Slot", (), {})()
            slot.index = i
            slot.pre_done, slot.post_done, slot.released = rt.event(), rt.event(), rt.event()
            slot.post_done.record()
            slot.released.record()
            slot.requests = [None] * self.microbatch  # (inputs, image_sizes) of the requests staged into the slot
            slot.fill, slot.enqueued, slot.generation = 0, True, 0
            slot.dirty_inputs = False  # host mirrors of the per-image scalars hold something `flush_inputs` has not shipped yet
            slot.seq = -1  # order of acquisition (submission order of the slot's requests)
            slot.collected = [False] * self.microbatch  # requests whose result the caller already holds
            slot.compute_stream = self.compute_streams[i % compute_streams]
            self.slots.append(slot)
        self._geometry = (Hp, Wp)
        self._build_plans()
        self._next = 0
        self._filling = None  # the slot requests are being staged into

    def _build_plans(self):
        """(Re)build every slot's launch plan and pair of graph halves for the model's current arithmetic, and replay them once."""
        from dd3d_amd.engine import ForwardPlan
        Hp, Wp = self._geometry
        rt = self.rt
        for slot in self.slots:
            extra = dict(device="cpu", dry_run=True) if rt.dry_run else {}
            # several slots in flight: tiled for CU-time, not for one launch's latency (engine.tiling.THROUGHPUT_TILE_TABLE; entries exist
            # for the plan geometries measured in situ, everything else falls back to the per-launch table)
            policy = "throughput" if (len(self.slots) > 1 and self.microbatch > 1) else None
            p = ForwardPlan(self.model, self.B * self.microbatch, Hp, Wp, world_size=self.world, rank=self.rank, exchange=self.exchange, tile_policy=policy, **extra)
            rt.warm(p)
            slot.plan = p
            slot.pre_graph = rt.capture(slot, p, 0, p.num_pre_nms_ops, "pre")
            slot.post_graph = rt.capture(slot, p, p.num_pre_nms_ops, None, "post")
        rt.synchronize()
        # first replay of every slot's graphs (a hipGraph's first launch uploads it: ~10x a steady replay), on the streams they will use;
        # a collective is NOT issued here (the ranks would have to agree on it) -- the post half runs on the zeroed record
        if not rt.dry_run:
            for slot in self.slots:
                with rt.on(slot.compute_stream):
                    slot.pre_graph.replay()
                    slot.pre_done.record(slot.compute_stream)
                self.post_stream.wait_event(slot.pre_done)
                with rt.on(self.post_stream):
                    slot.post_graph.replay()
                    slot.post_done.record(self.post_stream)
            rt.synchronize()
        for slot in self.slots:
            slot.plan.status.zero_()
        self.plan = self.slots[0].plan

    def _fall_back(self, err):
        """The range guard of the default f16x2 arithmetic fired on a request (DD3D.forward's behaviour, for the runner): on a model on the
        default arithmetic, drain the pipeline, rebuild every slot on the next arithmetic of engine.plan.relax_arithmetic (plane scale 16 -> 4 -> 1, then the
        three-term bf16 split) and re-run the requests that were in flight
        (their inputs are still referenced by their slots).  Several ranks: the verdict travels in the exchanged records
        (engine.ForwardPlan.check_status), so every rank raises for the same slot run and all of them rebuild and re-run together, issuing
        the same collectives in the same order -- provided every rank collects its results in the same order, as a data-parallel loop does."""
        from dd3d_amd.engine import relax_arithmetic
        for cs in self.compute_streams:
            cs.synchronize()
        self.post_stream.synchronize()
        if not relax_arithmetic(self.model, err):  # plane scale 16 -> 4 -> 1, then bf16x3; an explicitly chosen arithmetic raises
            raise err
        self._build_plans()
        # Only slots that still owe a result are re-run (a slot whose requests were all collected would only add a collective), and in
        # SUBMISSION order -- the order in which their collectives were first issued, which every rank shares (round-3 advisor: slot-index
        # order differs from it once the ring has wrapped).
        for slot in sorted(self.slots, key=lambda sl: sl.seq):
            staged = [(j, r) for j, r in enumerate(slot.requests[:slot.fill]) if r is not None]
            if slot.generation == 0 or not staged or all(slot.collected[j] for j, _ in staged):
                continue
            with self.rt.on(slot.compute_stream):
                for j, (inputs, _) in staged:
                    self.model.stage_inputs(inputs, plan=slot.plan, first=j * self.B, partial=True, flush=False)
            slot.dirty_inputs = True
            if slot.enqueued:
                self._enqueue(slot)

    def _enqueue(self, slot):
        cs, ps = slot.compute_stream, self.post_stream
        with self.rt.on(cs):
            if slot.dirty_inputs:  # sizes / intrinsics / resize targets of every request staged into the slot: one hand-over per slot run
                slot.plan.flush_inputs()
                slot.dirty_inputs = False
            slot.pre_graph.replay()
            slot.pre_done.record(cs)
        ps.wait_event(slot.pre_done)
        ps.wait_event(slot.released)
        with self.rt.on(ps):
            if self.exchange:
                gather_candidates(slot.plan.gather_pairs())
            slot.post_graph.replay()
            slot.plan.fetch()  # counts / status / range-guard words -> pinned memory, behind the post half (result() reads them there)
            slot.post_done.record(ps)
        slot.enqueued = True
        if self._filling is slot:
            self._filling = None

    def _acquire(self):
        slot = self.slots[self._next % len(self.slots)]
        self._next += 1
        # the slot's previous step must be over before its buffers are rewritten (a no-op wait when it finished long ago)
        slot.compute_stream.wait_event(slot.post_done)
        slot.compute_stream.wait_event(slot.released)
        slot.fill, slot.enqueued = 0, False
        slot.generation += 1
        slot.seq = self._next
        slot.requests = [None] * self.microbatch
        slot.collected = [False] * self.microbatch
        return slot

    def _position(self):
        """(slot, position) the next request goes to."""
        if self._filling is None:
            self._filling = self._acquire()
        slot = self._filling
        j = slot.fill
        slot.fill += 1
        return slot, j

    def step(self):
        """One request on inputs already resident in the slot's buffers (bench: `stage_all`); the slot is enqueued when its
        `microbatch`-th request arrives.  Returns the slot."""
        slot, j = self._position()
        if slot.fill == self.microbatch:
            self._enqueue(slot)
        return slot

    def flush(self):
        """Enqueue a partly filled slot (the whole plan runs; the unfilled positions' outputs are never read)."""
        if self._filling is not None and not self._filling.enqueued:
            self._enqueue(self._filling)

    def stage_all(self, batched_inputs):
        """Inputs resident before a timed region (bench): `batched_inputs` is one request's B inputs -- staged into every position of every
        slot -- or a list of `microbatch` requests: request j goes to position j of every slot."""
        per_pos = batched_inputs if isinstance(batched_inputs[0], (list, tuple)) else [batched_inputs] * self.microbatch
        assert len(per_pos) == self.microbatch
        for slot in self.slots:
            with self.rt.on(slot.compute_stream):
                for j, inp in enumerate(per_pos):
                    self.model.stage_inputs(inp, plan=slot.plan, first=j * self.B, partial=True, flush=False)
                slot.plan.flush_inputs()
        self.rt.synchronize()

    def submit(self, batched_inputs):
        slot, j = self._position()
        # device-resident inputs (DeviceInputMapper / DeviceResizer outputs) were produced on the caller's current stream: the staging
        # copies on the slot's compute stream must order after them, and the allocator must not recycle them before the copies ran
        slot.compute_stream.wait_stream(self.rt.current_stream())
        for x in batched_inputs:
            if x["image"].is_cuda:
                x["image"].record_stream(slot.compute_stream)
        with self.rt.on(slot.compute_stream):
            _, image_sizes = self.model.stage_inputs(batched_inputs, plan=slot.plan, first=j * self.B, partial=True, flush=False)
        slot.dirty_inputs = True
        slot.requests[j] = (batched_inputs, image_sizes)
        if slot.fill == self.microbatch:
            self._enqueue(slot)
        return (slot, j, slot.generation)

    def result(self, handle):
        slot, j, gen = handle if isinstance(handle, tuple) else (handle, 0, handle.generation)
        assert gen == slot.generation, "the slot of this request has been re-used: collect results within `depth` slots of submitting"
        if not slot.enqueued:
            self.flush()
        slot.post_done.synchronize()
        inputs, image_sizes = slot.requests[j]
        while True:
            try:
                out = self.model.collect(slot.plan, inputs, image_sizes, first=j * self.B)
                break
            except FloatingPointError as e:
                self._fall_back(e)  # (raises once nothing is left to relax; each round re-runs the requests in flight)
                slot.post_done.synchronize()
        slot.collected[j] = True
        slot.released.record()  # the copies out of the detection buffer are enqueued: later steps of this slot order after them
        return out

    def synchronize(self):
        self.flush()
        for cs in self.compute_streams:
            cs.synchronize()
        self.post_stream.synchronize()
