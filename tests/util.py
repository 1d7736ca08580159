"""Helpers shared by the parity tests (test infrastructure; may import the oracle)."""
import torch


_BUNDLES = {}


def bundle(experiment, tag, overrides=None):
    """(cfg, synthetic CPU state_dict) of one of the reference experiments, cached per session."""
    key = (experiment, tag, repr(overrides))
    if key not in _BUNDLES:
        import dd3d_amd.modeling  # noqa: F401
        from dd3d_amd import META_ARCH_REGISTRY, get_cfg
        from dd3d_amd.synthetic import load_calib, make_state_dict
        cfg = get_cfg(experiment, overrides)
        model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
        _BUNDLES[key] = (cfg, make_state_dict(model, calib=load_calib(tag)))
    return _BUNDLES[key]


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor*max|b|): relative error that does not blow up on near-zero entries."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if b.numel() == 0:
        return 0.0
    den = b.abs().clamp(min=floor * max(float(b.abs().max()), 1e-30))
    return float(((a - b).abs() / den).max())


def max_abs(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max()) if b.numel() else 0.0


def quat_err(qa, qb):
    """quaternions compared up to sign (pytorch3d's matrix_to_quaternion does not canonicalise it)."""
    qa, qb = qa.detach().float().cpu(), qb.detach().float().cpu()
    if qb.numel() == 0:
        return 0.0
    return float(torch.minimum((qa - qb).abs().amax(1), (qa + qb).abs().amax(1)).max())


def gpu_model(cfg, sd, use_graph=True, math=None):
    from dd3d_amd import build_model
    model = build_model(cfg)
    model.load_state_dict(sd)
    model.use_graph = use_graph
    model.math = math
    return model


def oracle_heads_to_plan(plan, st, num_classes):
    """Overwrite the plan's head-map buffers with the oracle's head maps (NCHW -> fused NHWC layout)."""
    for l in range(len(st["logits"])):
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
        plan.cls_maps[l].t.zero_()
        plan.cls_maps[l].t[..., :num_classes] = nhwc(st["logits"][l]).to(plan.device)
        if "attr" in st:  # nuScenes extras ride on the cls map: [logits | attr logits | speed]
            na = st["attr"][l].shape[1]
            plan.cls_maps[l].t[..., num_classes:num_classes + na] = nhwc(st["attr"][l]).to(plan.device)
            plan.cls_maps[l].t[..., num_classes + na:num_classes + na + 1] = nhwc(st["speed"][l]).to(plan.device)
        plan.b2d_maps[l].t.zero_()
        plan.b2d_maps[l].t[..., 0:4] = nhwc(st["box2d_reg"][l]).to(plan.device)
        plan.b2d_maps[l].t[..., 4:5] = nhwc(st["centerness"][l]).to(plan.device)
        fused = torch.cat([st["quat"][l], st["ctr"][l], st["depth"][l], st["size"][l], st["conf"][l]], 1)
        plan.b3d_maps[l].t.zero_()
        plan.b3d_maps[l].t[..., :fused.shape[1]] = nhwc(fused).to(plan.device)


def candidates_from_plan(plan, b=0):
    """Decoded candidates of image b as a dict of CPU tensors, concatenated over levels (Instances.cat order)."""
    import numpy as np
    cand = plan.cand[b].cpu()
    counts = plan.counts[b].cpu().tolist()
    slots = torch.cat([torch.arange(c) + plan.slot_off[l] for l, c in enumerate(counts)]).long()
    c = cand[:, slots]
    ints = lambda row: torch.from_numpy(c[row].numpy().view(np.int32).copy()).long()
    return dict(
        pred_boxes=c[0:4].T, scores=c[4], scores_3d=c[5], pred_classes=ints(6), flat_index=ints(7), locations=c[8:10].T, quat=c[10:14].T,
        proj_ctr=c[14:16].T, depth=c[16:17].T, size=c[17:20].T, counts=counts,
        fpn_levels=torch.cat([torch.full((c_, ), l) for l, c_ in enumerate(counts)]).long()
    )


def candidate_margins(plan, st, cfg, b=0, ref_b=None):
    """End-to-end candidate membership of image b: HIP (plan.cand, from its own head maps) vs oracle (st; image `ref_b` of the oracle's
    batch, default b -- the bench compares position j of a slot's plan with a one-image oracle forward).  Selection is a hard
    threshold on sigma(cls)*sigma(ctr) (fcos2d.py:280-283) followed by a per-level top-k (fcos2d.py:309-317), so a ~1e-7 relative
    difference in a logit may flip a candidate that sits ON a cut.  Returns (n_hip, n_ref, margins): `margins` holds, for every
    candidate only one side selected, the distance of the ORACLE's score from the cut that decided it (PRE_NMS_THRESH, or the
    k-th largest score of the level when the top-k cut was active).  Parity demands every such distance to be ~0."""
    inf = cfg.DD3D.FCOS2D.INFERENCE
    C, topk, thr = int(cfg.DD3D.NUM_CLASSES), int(inf.PRE_NMS_TOPK), float(inf.PRE_NMS_THRESH)
    c = candidates_from_plan(plan, b)
    b = b if ref_b is None else ref_b  # (from here on: the oracle's image)
    hip_keys = set(zip(c["fpn_levels"].tolist(), c["flat_index"].tolist()))
    ref_keys, margins = set(), []
    dense = []
    for l, info in enumerate(st["level_info"]):
        fg, cl, tk = info[b]["fg_inds"], info[b]["class_inds"], info[b]["topk_indices"]
        e = fg * C + cl
        e = e[tk] if tk is not None else e
        ref_keys |= {(l, int(v)) for v in e.tolist()}
        s = st["logits"][l][b].permute(1, 2, 0).reshape(-1, C).sigmoid()
        ctr = st["centerness"][l][b].permute(1, 2, 0).reshape(-1, 1).sigmoid()
        s_thr = s * ctr if bool(inf.THRESH_WITH_CTR) else s
        rank = (s * ctr).reshape(-1)
        n_pass = int((s_thr > thr).sum())
        kth = float(rank[(s_thr > thr).reshape(-1)].topk(topk).values.min()) if n_pass > topk else None
        dense.append((s_thr.reshape(-1), rank, kth))
    for (l, flat) in hip_keys ^ ref_keys:
        s_thr, rank, kth = dense[l]
        d = abs(float(s_thr[flat]) - thr)
        if kth is not None:
            d = min(d, abs(float(rank[flat]) - kth))
        margins.append(d)
    return len(hip_keys), len(ref_keys), margins
