"""Dev tool: time and HBM traffic of the dominant convolution kernels of ANY configuration (the tower probe of tests/gpu_pmc_probe.py
covers DD3D-DLA34 only).

    python tests/gpu_kernel_traffic.py run <exp> <B> <H> <W> <iters> <out.json>
        builds the plan, writes per kernel instantiation what ONE forward asks of it (launches, FLOP, algorithmic bytes = input planes
        + filter planes + outputs + residual reads, each counted once per launch) to <out.json>, then runs the plan `iters` times
        op by op (no hipGraph) -- the command rocprofv3 wraps, once with --kernel-trace --stats, once per --pmc pass.
    python tests/gpu_kernel_traffic.py report <out.json> <kernel_stats.csv> <fetch counter csv> <write counter csv> <summary.json>
        joins the three: per kernel average launch time, f32-equivalent TFLOP/s and fraction of the mode's roofline, HBM bytes per
        launch (FETCH_SIZE x 2 KiB, WRITE_SIZE KiB: MI355X_MICROARCH.md, HBM section) against the algorithmic bytes, L2 hit rate.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PRODUCTS = {1: 6, 2: 3, 3: 1, 4: 3}  # 16-bit matrix products per f32 product (bf16x3, bf16x2, bf16, f16x2)


def run(exp, B, H, W, iters, out):
    import torch
    import __graft_entry__ as g
    g.build()
    from dd3d_amd import build_model, get_cfg, hip
    from dd3d_amd.engine import ConvOp, kernel_signature
    from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
    cfg = get_cfg(exp)
    model = build_model(cfg)
    tag = ("dla34" if "dla34" in exp else "v99") + ("_nusc" if "nusc" in exp else "_kitti")
    model.load_state_dict(make_state_dict(model, calib=load_calib(tag)))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, H, W, dataset="nusc" if "nusc" in exp else "kitti"))
    table = {}
    for op in plan.ops:
        if not isinstance(op, ConvOp):
            continue
        L = op.L
        npl = hip.MATH_PLANES[op.math] if op.in_planes else 0
        byt = 0
        for s, (wf, wp) in zip(op.desc["segs"], op.out_forms):
            vin, vout = s["in"], s["out"]
            m_in, m_out = vin.B * vin.H * vin.W, vout.B * vout.H * vout.W
            byt += m_in * L.Cin * (npl * 2 if op.in_planes else 4)
            byt += m_out * L.N * ((4 if wf else 0) + (hip.MATH_PLANES[op.math] * 2 if wp else 0))
            if s.get("res") is not None:
                byt += m_out * L.N * 4
        # one filter set per distinct weight tensor of the launch (the towers share theirs over the levels)
        nfilters = len({id(s["w"]) for s in op.desc["segs"]})
        byt += nfilters * L.Npad * L.Kpad * (npl * 2 if op.in_planes else 4)
        e = table.setdefault(kernel_signature(op), dict(launches=0, flop=0, algorithmic_bytes=0, math=int(op.math), ops=[]))
        e["launches"] += 1
        e["flop"] += 2 * op.macs
        e["algorithmic_bytes"] += byt
        e["ops"].append(op.name)
    json.dump(dict(exp=exp, B=B, H=H, W=W, iters=iters, kernels=table), open(out, "w"), indent=1)
    for _ in range(iters):
        plan.run()
    torch.cuda.synchronize()
    plan.check_status()


def _counter_rows(path):
    rows = {}
    files = [path] if os.path.isfile(path) else [os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.endswith("counter_collection.csv")]
    for f in files:
        for row in csv.DictReader(open(f)):
            rows.setdefault((row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    return rows


def report(plan_json, stats_csv, fetch_path, write_path, out):
    plan = json.load(open(plan_json))
    stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(stats_csv))}
    fetch, write = _counter_rows(fetch_path), _counter_rows(write_path)
    summary = dict(config=f"{plan['exp']} B={plan['B']} {plan['H']}x{plan['W']}", forwards_profiled=plan["iters"], kernels={},
                   method="rocprofv3 --kernel-trace --stats (time), --pmc FETCH_SIZE (own pass), --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (own pass); "
                          "FETCH_SIZE x 2 x 1024 B (gfx950 reports half the bytes of wide reads), WRITE_SIZE x 1024 B; per-launch averages over "
                          "every launch of the instantiation in the forward (mixed layer shapes), against the same average of algorithmic bytes")
    for sig, e in sorted(plan["kernels"].items(), key=lambda kv: -float(stats.get(kv[0], {}).get("TotalDurationNs", 0))):
        st = stats.get(sig)
        if st is None:
            continue
        n = e["launches"]
        avg_us = float(st["AverageNs"]) / 1e3
        flop, alg = e["flop"] / n, e["algorithmic_bytes"] / n
        peak = 2500.0 / PRODUCTS.get(e["math"], 1) if e["math"] else 157.3
        k = dict(launches_per_forward=n, share_of_gpu_time_pct=float(st["Percentage"]), avg_launch_us=round(avg_us, 2),
                 tflops_f32_equiv=round(flop / avg_us / 1e6, 1), roofline_tflops=round(peak, 1), frac=round(flop / avg_us / 1e6 / peak, 3),
                 algorithmic_bytes_per_launch=int(alg))
        f, w = fetch.get((sig, "FETCH_SIZE")), write.get((sig, "WRITE_SIZE"))
        if f and w:
            hbm = sum(f) / len(f) * 2 * 1024 + sum(w) / len(w) * 1024
            k.update(hbm_bytes_per_launch=int(hbm), ratio_to_algorithmic=round(hbm / alg, 3), achieved_hbm_gbps=round(hbm / avg_us / 1e3, 1))
        h, m = write.get((sig, "TCC_HIT_sum")), write.get((sig, "TCC_MISS_sum"))
        if h and m:
            k["tcc_hit_rate"] = round(sum(h) / (sum(h) + sum(m)), 4)
        summary["kernels"][sig] = k
    json.dump(summary, open(out, "w"), indent=1)
    for sig, k in list(summary["kernels"].items())[:6]:
        print(sig, json.dumps(k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
    else:
        report(*sys.argv[2:7])
