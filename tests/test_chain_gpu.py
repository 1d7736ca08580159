"""Chain launches (include/dd3d_hip.h, dd3d_conv_launch.chain; csrc/conv_planes_row.hip CHAIN instantiations; engine.PlanBase.merge_chains):
dependent 3 x 3 convolutions in ONE launch must produce EXACTLY what one launch per convolution produces -- same kernels' arithmetic, same
tiles, same split-K order -- for every backbone level, the head towers, with and without split-K, with residuals read from earlier segments;
and they must keep doing so under load (several slots in flight, warm caches), which is where a stale hand-over between XCDs would show."""
import pytest
import torch

from tests.util import bundle, gpu_model

pytestmark = pytest.mark.gpu


def _forward(cfg, sd, inputs, chain, monkeypatch, graph=True, math=None):
    monkeypatch.setenv("DD3D_CHAIN", chain)
    model = gpu_model(cfg, sd, use_graph=graph, math=math)
    plan, sizes = model.stage_inputs(inputs)
    for _ in range(3):  # the third forward runs on warm caches and counters that earlier launches have used and cleared
        plan.run()
    torch.cuda.synchronize()
    plan.check_status()
    return model, plan, sizes


def _same_plan_state(pc, pf):
    """Every activation buffer (both storages), every head map and the detections of the two plans, bit for bit."""
    assert sorted(pc.bufs) == sorted(pf.bufs)
    for name, bc in pc.bufs.items():
        bf = pf.bufs[name]
        if bc.t is not None and bf.t is not None:
            assert torch.equal(bc.t, bf.t), name
        if bc.p is not None:
            assert bf.p is not None and torch.equal(bc.p, bf.p), name  # the split planes: what the convolutions hand to each other
    assert torch.equal(pc.det_count, pf.det_count) and torch.equal(pc.det, pf.det)
    assert torch.equal(pc.cand, pf.cand) and torch.equal(pc.counts, pf.counts)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exp,tag,B,H,W", [("dd3d_kitti_dla34", "dla34_kitti", 2, 128, 256), ("dd3d_kitti_dla34", "dla34_kitti", 1, 384, 1280),
                                           ("dd3d_kitti_dla34", "dla34_kitti", 4, 384, 1280), ("dd3d_kitti_v99", "v99_kitti", 2, 128, 256),
                                           ("dd3d_nusc_dla34", "dla34_nusc", 6, 128, 224)],
                         ids=["dla34_small_b2", "dla34_kitti_b1", "dla34_kitti_b4", "v99_small_b2", "dla34_nusc_b6"])
def test_chain_launches_equal_one_launch_per_convolution(hiplib, monkeypatch, exp, tag, B, H, W):
    from dd3d_amd.engine import ConvOp
    from dd3d_amd.synthetic import make_inputs
    cfg, sd = bundle(exp, tag)
    inputs = make_inputs(B, H, W, dataset="nusc" if "nusc" in exp else "kitti")
    mc, pc, _ = _forward(cfg, sd, inputs, "1", monkeypatch)
    mf, pf, _ = _forward(cfg, sd, inputs, "0", monkeypatch)
    chains = [op for op in pc.ops if isinstance(op, ConvOp) and op.chain]
    assert chains and not any(isinstance(op, ConvOp) and op.chain for op in pf.ops)
    assert len(pf.ops) - len(pc.ops) == sum(len(op.parts) - 1 for op in chains)
    print(f"[chain] {exp} B={B} {H}x{W}: {len(pf.ops)} -> {len(pc.ops)} launches; chains: " +
          ", ".join(f"{len(op.parts)} x {op.info['tile_name']}" + (f" split-K {op.info['splitk']}" if op.info["splitk"] > 1 else "") for op in chains))
    _same_plan_state(pc, pf)
    for op in chains:  # the counters are zero again
        assert int(op.chain_sync.abs().sum()) == 0, op.name


@pytest.mark.timeout(900)
@pytest.mark.parametrize("math", ["bf16x3", "bf16x2", "bf16"])
def test_chain_launches_in_the_other_split_operand_modes(hiplib, monkeypatch, math):
    """The CHAIN instantiations exist for every split-plane arithmetic (three planes for bf16x3: another staging layout of the write-through
    stores, six 16-byte residual pieces): the same bit-for-bit comparison on the small DLA-34 case."""
    from dd3d_amd.engine import ConvOp
    from dd3d_amd.synthetic import make_inputs
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti")
    inputs = make_inputs(2, 128, 256)
    mc, pc, _ = _forward(cfg, sd, inputs, "1", monkeypatch, math=math)
    mf, pf, _ = _forward(cfg, sd, inputs, "0", monkeypatch, math=math)
    assert any(isinstance(op, ConvOp) and op.chain for op in pc.ops) and not any(isinstance(op, ConvOp) and op.chain for op in pf.ops)
    _same_plan_state(pc, pf)


@pytest.mark.timeout(900)
def test_chain_launches_under_load(hiplib, monkeypatch):
    """Five slots in flight, four distinct images each, forty rounds: every slot run must reproduce the detections of the plan without chain
    launches -- the hand-over between dependent tiles (write-through stores, agent-coherent loads, counters) holds with warm caches, other
    streams' kernels sharing the CUs and the L2s, and uneven arrival of the producers."""
    from dd3d_amd.parallel import PipelinedForward
    from dd3d_amd.synthetic import make_inputs
    cfg, sd = bundle("dd3d_kitti_dla34", "dla34_kitti")
    reqs = [make_inputs(1, 384, 1280, seed=4000 + j) for j in range(4)]
    monkeypatch.setenv("DD3D_CHAIN", "0")
    mf = gpu_model(cfg, sd, use_graph=True)
    pf, _ = mf.stage_inputs([r[0] for r in reqs])
    pf.run()
    torch.cuda.synchronize()
    want_n, want = pf.det_count.clone(), pf.det.clone()
    monkeypatch.setenv("DD3D_CHAIN", "1")
    model = gpu_model(cfg, sd, use_graph=True)
    runner = PipelinedForward(model, 1, 384, 1280, depth=5, compute_streams=5, microbatch=4)
    runner.stage_all(reqs)
    noise = torch.zeros((64, 1024, 1024), device="cuda")  # another stream's traffic through the L2s / the fabric
    side = torch.cuda.Stream()
    for rnd in range(40):
        for _ in range(20):
            runner.step()
        if rnd % 3 == 0:
            with torch.cuda.stream(side):
                noise.add_(1.0)
        runner.synchronize()
        for sl in runner.slots:
            sl.plan.check_status()
            assert torch.equal(sl.plan.det_count, want_n), (rnd, sl.index)
            assert torch.equal(sl.plan.det, want), (rnd, sl.index)
