"""Which buffers of a plan with chain launches differ from the one-launch-per-convolution plan, and where (development tool).

    python tests/gpu_chain_probe.py [B] [H] [W] [runs]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from dd3d_amd import build_model, get_cfg  # noqa: E402
from dd3d_amd.engine import ConvOp  # noqa: E402
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict  # noqa: E402


def plan_for(chain, B, H, W, runs):
    os.environ["DD3D_CHAIN"] = chain
    cfg = get_cfg("dd3d_kitti_dla34")
    model = build_model(cfg)
    model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti")))
    model.use_graph = False
    plan, _ = model.stage_inputs(make_inputs(B, H, W))
    for _ in range(runs):
        plan.run()
    torch.cuda.synchronize()
    try:
        plan.check_status()
    except Exception as e:
        print("status:", e)
    return model, plan


def main():
    B, H, W, runs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 2), (2, 128), (3, 256), (4, 1)))
    mf, pf = plan_for("0", B, H, W, runs)
    mc, pc = plan_for("1", B, H, W, runs)
    for op in pc.ops:
        if isinstance(op, ConvOp) and op.chain:
            print("chain", op.name, op.info["tile_name"], "splitk", op.info["splitk"], "blocks", op.info["blocks"], "sync", int(op.chain_sync.abs().sum()))
    for name, bc in pc.bufs.items():
        bf = pf.bufs[name]
        if bc.p is None:
            continue
        d = (bc.p != bf.p)
        if bool(d.any()):
            bad = d.any(3).any(2)  # [chunk][pixel]
            pix = bad.any(0).nonzero().flatten()
            ch = bad.any(1).nonzero().flatten()
            print(f"DIFF {name}: {int(bad.sum())} (chunk, pixel) cells of {bad.numel()}; chunks {ch.tolist()[:12]}; pixels {pix[:6].tolist()} ... {pix[-6:].tolist()} ({pix.numel()} pixels)")
        else:
            print(f"same {name}")
    print("detections equal:", bool(torch.equal(pc.det, pf.det)))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("step", "indep")):
    main()


def stepwise():
    """Both plans launch by launch: after which launch does which buffer first differ?"""
    from dd3d_amd import hip
    B, H, W = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1), (3, 384), (4, 1280)))
    mf, pf = plan_for("0", B, H, W, 1)
    mc, pc = plan_for("1", B, H, W, 1)
    st = hip.current_stream()
    names_f = [op.name for op in pf.ops]
    i_f = 0
    reported = set()
    sync_each = os.environ.get("PROBE_SYNC", "1") == "1"
    for op in pc.ops:
        parts = getattr(op, "parts", None) or [op.name]
        op(pc.lib, st)
        outs = []
        for part in parts:
            assert names_f[i_f] == part, (names_f[i_f], part)
            pf.ops[i_f](pf.lib, st)
            d = getattr(pf.ops[i_f], "desc", None) or {}
            outs += [sg["out"].buf.name for sg in d.get("segs", [])] + [d[k].buf.name for k in ("vout", "dst", "fine", "out") if k in d and hasattr(d[k], "buf")]
            i_f += 1
        if sync_each:
            torch.cuda.synchronize()
        else:
            continue
        for name in dict.fromkeys(outs):
            bc, bf = pc.bufs[name], pf.bufs[name]
            if bc.p is None:
                continue
            d = (bc.p != bf.p)
            if bool(d.any()):
                bad = d.any(3).any(2)
                pix = bad.any(0).nonzero().flatten()
                reported.add(name)
                print(f"after {op.name}: {name} differs in {pix.numel()} pixels: {pix[:8].tolist()} ... {pix[-4:].tolist()}; chunks {bad.any(1).nonzero().flatten().tolist()}")
                c0, p0 = [int(v) for v in bad.nonzero()[0]]
                hc, hf = bc.p[c0, p0].view(torch.float16).float(), bf.p[c0, p0].view(torch.float16).float()
                print("   e.g. chunk", c0, "pixel", p0, "chain hi", hc[0, :6].tolist(), "lo", hc[1, :6].tolist(), "| flat hi", hf[0, :6].tolist(), "lo", hf[1, :6].tolist())
                runs, prev, start = [], None, None
                for v in pix.tolist():
                    if prev is None or v != prev + 1:
                        if prev is not None:
                            runs.append((start, prev))
                        start = v
                    prev = v
                runs.append((start, prev))
                print("   pixel runs:", runs[:24], "..." if len(runs) > 24 else "")
            else:
                print(f"after {op.name}: {name} same")
    if not sync_each:
        torch.cuda.synchronize()
        for name, bc in pc.bufs.items():
            if bc.p is not None and bool((bc.p != pf.bufs[name].p).any()):
                reported.add(name)
                print("no sync between launches:", name, "differs")
    print("done;", len(reported), "buffers differ")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "step":
    stepwise()


def independent():
    """A chain launch whose segments do NOT depend on each other: the same convolution twice (second copy into a scratch buffer), with and
    without a residual -- is the CHAIN instantiation itself exact?"""
    from dd3d_amd import hip
    B, H, W = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1), (3, 384), (4, 1280)))
    mf, pf = plan_for("0", B, H, W, 1)
    st = hip.current_stream()
    for name in ("level2.tree1.conv2", "level2.tree2.conv1", "level3.tree1.tree1.conv2", "level4.tree1.tree2.conv1", "level4.tree2.tree1.conv2"):
        op = next(o for o in pf.ops if o.name == name)
        sg = op.desc["segs"][0]
        want = sg["out"].buf.p.clone()
        c = op.ctor
        outs = [pf.buf(f"probe.{name}.{k}", sg["out"].B, sg["out"].H, sg["out"].W, sg["out"].C, kind="planes") for k in range(3)]
        segs = [dict(sg, out=o.view(), dep=None) for o in outs]
        ch = ConvOp(pf, c["meta"], 1, 1, segs, c["relu"], tile=c["tile"], splitk=c["splitk"], name="probe." + name, math=c["math"], chain=True)
        for rep in range(3):
            ch(pf.lib, st)
            torch.cuda.synchronize()
            c0 = sg["out"].c0 // 32
            n = sg["out"].C // 32
            res = []
            for o in outs:
                d = (o.p != want[c0:c0 + n])
                res.append(int(d.any(3).any(2).any(0).sum()))
            print(f"{name} (res {op.res_forms}, tile {op.info['tile_name']}, split-K {op.info['splitk']}) rep {rep}: differing pixels per copy {res}; sync {int(ch.chain_sync.abs().sum())}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "indep":
    independent()
