"""N > 1 code path on the GPU: two processes share the one GPU of the test box; gloo (host-staged) carries the candidate gather,
everything else -- the two captured hipGraph halves, NMS over all W*B images, rank-offset collect -- is the product path.  The RCCL
transport itself is exercised with a single-rank "nccl" group between the same two graph halves."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_match_single_rank(hiplib):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_dist_check.py"), "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "dist check: {" in r.stdout and "False" not in r.stdout.split("dist check:")[-1]


def test_camera_sharded_sample_on_two_ranks_matches_single_rank(hiplib):
    """The reader of the exchange: a nuScenes sample's cameras split 3 / 3 over two ranks, aggregated by the sample's owner out of the
    gathered records (2D NMS of all six cameras + the sample-level BEV NMS), bit-identical to the single-rank forward of the sample."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_dist_check.py"), "cameras"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "camera-sharded check: {" in r.stdout and "False" not in r.stdout.split("camera-sharded check:")[-1]


def test_range_guard_verdict_is_shared_by_all_ranks(hiplib):
    """Only rank 1's activations leave the half range; both ranks must fall back to bf16x3 on the same step (the verdict rides in the
    exchanged records), in DistributedForward and in PipelinedForward, and return the bf16x3 model's detections."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_dist_check.py"), "guard"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]
    assert "range-guard consensus check: {" in r.stdout and "False" not in r.stdout.split("range-guard consensus check:")[-1]


def test_rccl_transport_single_rank(hiplib):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_rccl_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl check: ok=True" in r.stdout


def test_rccl_all_gather_captured_inside_the_step_graph(hiplib):
    """DD3D_GRAPH_EXCHANGE=1: pre half, RCCL all_gather and post half replayed as ONE hipGraph (single rank: all this box can validate)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_rccl_check.py"), "graph"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl check (all_gather inside the graph): ok=True" in r.stdout


def test_graph_exchange_probe_decides_on_the_transport(hiplib):
    """DD3D_GRAPH_EXCHANGE=probe (round-4 verdict item 6d): a tiny captured all_gather on a communicator of its own decides whether the
    step's collective is captured inside its hipGraph; whatever it decides, the step's detections equal the single-graph forward."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_rccl_check.py"), "probe"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "probe: captured all_gather usable =" in r.stdout and "ok=True" in r.stdout


@pytest.mark.parametrize("mode", ["", "nccl", "streams", "microbatch", "fallback"])
def test_pipelined_forward_equals_stepwise(hiplib, mode):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tests", "gpu_pipeline_check.py")] + ([mode] if mode else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok=True" in r.stdout


@pytest.mark.parametrize("pipeline", ["default", "0"])
def test_bench_multi_gpu_command_line(hiplib, pipeline):
    """The driver's N > 1 command, literally -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...` -- with two
    ranks sharing the one GPU of the test box over the host-staged gloo test transport (round-3 verdict: the first 8-GPU run must not be
    the first execution of bench.py's N > 1 lines).  Checks the JSON line the driver parses: whole-job value, n_gpus, the parallelism
    string, no pipeline error -- and that the line says the transport was NOT RCCL."""
    import json
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DD3D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--repeat-blocks", "1"]
    if pipeline != "default":
        cmd += ["--pipeline", pipeline]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 1 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]  # whole-job images / s = ranks x batch / step time
    c = d["config"]
    assert c["parallelism"] == "dp2+rccl_allgather_candidates" and c["global_batch"] == 2 and c["pipeline_error"] is None
    assert c["pipeline_slots"] == (0 if pipeline == "0" else 5)
    assert "NOT RCCL" in c["transport"] and "gloo" in c["transport"]
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    # round 5: what lets a reader of the line confirm the transport carried N ranks on N devices, and that no rank idled
    assert c["rccl_nranks"] == 0 and c["exchange_selftest"]["nranks"] == 2 and c["exchange_selftest"]["backend"] == "gloo"  # (0: this run was NOT on RCCL)
    assert len(c["rank_devices"]) == 2 and all(dv["device"] == 0 and dv["pci_bus_id"] for dv in c["rank_devices"])
    assert len(c["rank_ms_per_step"]) == 2 and max(c["rank_ms_per_step"]) <= d["ms_per_step"] * 1.001 and c["graph_exchange"] is False
    # (the default issue mode stages four DISTINCT images per slot -- one per request position; `--pipeline 0` one)
    assert d["work_verified"]["identical_across_slots"] and d["work_verified"]["detections_per_image"] == [100] * (1 if pipeline == "0" else 4)


def test_bench_single_gpu_line_carries_parity_and_baselines(hiplib):
    """The driver's N = 1 command (fewer CPU forwards): the ONE JSON line must carry the metric, the roofline object, the CPU baseline,
    the parity object (the metric's second half: 3D-box L1 vs the oracle, bars met) and the read-back of the timed work."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-forwards", "2", "--repeat-blocks", "1",
           "--e2e-requests", "24"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "images/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    rf, cb, pr = d["roofline"], d["cpu_baseline"], d["parity"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"] and rf["unit"] == "TFLOP/s"
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    # parity is that of the TIMED launch plan: slot 0's four-image plan, every request position the CPU leg ran an oracle forward for
    assert pr["pass"] and pr["images_compared"] == 2 and pr["detections_hip"] == pr["detections_oracle"] == 200 and pr["off_cut_flips"] == 0
    assert "4-image plan" in pr["image"] and pr["matched"] >= 198
    assert pr["corners_l1_rel"] <= 1e-3 and pr["box3d_l1_tvec_size_rel"] <= 1e-3 and pr["box2d_rel_max"] <= 1e-3
    assert pr["int_mismatches"] == pr["rank_swaps"] or pr["on_cut_flips"] > 0  # same detections at the same ranks, up to score ties / cuts
    assert d["work_verified"]["slot_positions_checked"] == 20 and d["work_verified"]["distinct_images_per_slot"] == 4
    assert d["config"]["f16x2_range"]["overflow_headroom_x"] > 4
    assert d["config"]["bs1_images_per_s"] > 0 and "DLA34" in d["config"]["workload"]
    # the end-to-end leg (distinct host images through submit() / result(): core.py:65 ... :153-164) and the second issue geometry
    e2e = d["config"]["e2e"]
    for src in ("pinned", "pageable"):
        assert e2e[src]["images_per_s"] > 0 and e2e[src]["requests"] == 24 and e2e[src]["detections_returned"] > 0
        assert e2e[src]["host_us_per_request_stage_inputs"] > 0 and e2e[src]["host_us_per_request_collect"] > 0
    assert e2e["pinned"]["images_per_s"] > 0.8 * d["value"]
    assert d["config"]["alt_issue"]["median_images_per_s"] > 0 and "error" not in d["config"]["alt_issue"]
