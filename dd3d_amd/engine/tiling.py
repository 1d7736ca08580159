"""Tile / split-K selection of the implicit-GEMM launches: the instantiated tiles per arithmetic mode, the measured tables
(dd3d_amd/data/tile_table_*.json), the analytic fallback, and the kernel names rocprofv3 prints (dd3d_amd.engine)."""
import ctypes as C
import math
import os

from collections import OrderedDict

import numpy as np
import torch

from dd3d_amd import hip
from dd3d_amd.layers import fold_norm

NUM_CU = 256  # MI355X


MATH_TILES = {  # tile configurations instantiated per arithmetic mode
    hip.MATH_F32: (hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x64, hip.TILE_128x32, hip.TILE_64x128),
    hip.MATH_BF16X3: (hip.TILE_256x128, hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x128, hip.TILE_128x128_W4, hip.TILE_64x64_W4,
                      hip.TILE_128x64_W4, hip.TILE_128x64_K2, hip.TILE_64x128_K2, hip.TILE_64x64_W4K2),
}
# the split-plane kernel (csrc/conv_planes.hip): one barrier per K-tile for every tile, so no "two K-tiles per barrier" variants
PLANE_TILES = (hip.TILE_256x128, hip.TILE_128x128, hip.TILE_128x64, hip.TILE_64x128, hip.TILE_128x128_W4, hip.TILE_64x64_W4, hip.TILE_128x64_W4,
               hip.TILE_256x128_T42, hip.TILE_128x256_T24, hip.TILE_256x256_W8, hip.TILE_128x32_W4, hip.TILE_192x256_W8)
ROW_ONLY_TILES = (hip.TILE_192x256_W8,)  # instantiated for the row-shared 3 x 3 kernel only
BIG_WAVE_TILES = (hip.TILE_256x128_T42, hip.TILE_128x256_T24, hip.TILE_256x256_W8, hip.TILE_192x256_W8)  # 8 accumulator blocks per wave; picked by the measured table only
PLANE_TILE_ALIAS = {hip.TILE_128x64_K2: hip.TILE_128x64, hip.TILE_64x128_K2: hip.TILE_64x128, hip.TILE_64x64_W4K2: hip.TILE_64x64_W4}
for _m in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2):
    MATH_TILES[_m] = PLANE_TILES
# blocks of a configuration that can share a CU (LDS-limited); the f32 kernels were measured, see profiles/
BLOCKS_PER_CU = {hip.TILE_128x128_W4: 2, hip.TILE_64x64_W4: 2, hip.TILE_128x64_W4: 2}  # tiles the split-bf16 kernel is instantiated for


MATH_NAMES = {"f32": hip.MATH_F32, "bf16x3": hip.MATH_BF16X3, "bf16x2": hip.MATH_BF16X2, "bf16": hip.MATH_BF16, "f16x2": hip.MATH_F16X2}


def default_math():
    """Arithmetic of the Cin % 32 == 0 convolutions (all accumulate in f32).  Env DD3D_MATH or model.math.
      f32-equivalent (measured against a float64 convolution they sit at the same ~3e-7 as exact f32, tests/test_conv_planes_gpu.py):
        "f16x2"  (default) two IEEE-half terms per operand, 3 cross products on the f16 matrix pipe; needs |activation| <= 65504 /
                 plane scale -- a kernel-side status word trips otherwise and the forward raises / falls back to "bf16x3"
        "bf16x3" three bf16 terms, 6 cross products; the full f32 exponent range
        "f32"    v_mfma_f32_32x32x2_f32, bitwise an fmaf chain (1/16 of the bf16 rate)
      reduced (what BASELINE.json's bf16 configurations name):
        "bf16x2" two bf16 terms, 3 products (~1e-5 relative);  "bf16" plain bf16 operands (~1e-2: misses the 1e-3 parity bar)"""
    return MATH_NAMES[os.environ.get("DD3D_MATH", "f16x2")]


# (TM, TN, WM, WN) of the split-plane kernels' tiles: 32 x 32 accumulator blocks per wave and the block's wave grid
TILE_WAVE_GRID = {hip.TILE_256x128: (2, 2, 4, 2), hip.TILE_128x128: (2, 1, 2, 4), hip.TILE_128x64: (1, 1, 4, 2), hip.TILE_64x128: (1, 1, 2, 4),
                  hip.TILE_128x128_W4: (2, 2, 2, 2), hip.TILE_64x64_W4: (1, 1, 2, 2), hip.TILE_128x64_W4: (2, 1, 2, 2),
                  hip.TILE_256x128_T42: (4, 2, 2, 2), hip.TILE_128x256_T24: (2, 4, 2, 2), hip.TILE_256x256_W8: (4, 2, 2, 4),
                  hip.TILE_128x32_W4: (1, 1, 4, 1), hip.TILE_192x256_W8: (3, 2, 2, 4)}


def kernel_signature(op):
    """Name of the kernel instantiation a ConvOp launches, as rocprofv3 prints it (bench.py / profiles bookkeeping)."""
    cfg = op.L.tile_cfg
    tm_tn_wm_wn = TILE_WAVE_GRID
    sk = "true" if op.L.splitk > 1 else "false"
    if op.in_planes:
        tm, tn, wm, wn = tm_tn_wm_wn[PLANE_TILE_ALIAS.get(cfg, cfg)]
        np_ = hip.MATH_PLANES[op.math]
        bm, bn = tm * 32 * wm, tn * 32 * wn
        L = op.L
        nk = L.Kpad // 32
        row = (L.KH == 3 and L.KW == 3 and L.stride == 1 and L.pad == 1 and (L.splitk == 1 or -(-nk // L.splitk) % 3 == 0)
               and os.environ.get("DD3D_CONV_ROW", "1") != "0")
        if row:  # csrc/conv_planes_row.hip: the three taps of a filter row share one A stage
            # ring depths: what the LIBRARY instantiates (dd3d_conv_row_rings: they are build-time properties of the .so); the formula below
            # (csrc/conv_planes_row.hip::RowRings with the product build's defaults) only serves a box without the library
            nsb = nsa = None
            try:
                b_, a_ = C.c_int32(), C.c_int32()
                if hip.lib().dd3d_conv_row_rings(PLANE_TILE_ALIAS.get(cfg, cfg), op.math, C.byref(b_), C.byref(a_)) == 0:
                    nsb, nsa = b_.value, a_.value
            except (hip.HipLibraryMissing, OSError):
                pass
            if nsb is None:
                nsb, nsa = row_rings_default(np_, bm, bn, wm * wn)
            chain = "true" if getattr(op, "chain", False) else "false"  # (CHAIN: dependent segments in one launch, dd3d_conv_launch.chain)
            return f"dd3d::conv_igemm_planes_row_kernel<{tm}, {tn}, {wm}, {wn}, {nsb}, {op.math}, {sk}, {nsa}, {chain}>"
        stage = np_ * (bm + bn) * 64
        ns = max(2, min(4, ((144 if (wm * wn == 8 or stage > 32768) else 72) * 1024) // stage))
        return f"dd3d::conv_igemm_planes_kernel<{tm}, {tn}, {wm}, {wn}, {ns}, {op.math}, {sk}, 0>"
    if op.math == hip.MATH_BF16X3:
        return f"dd3d::conv_igemm_bf16x3_kernel ({hip.TILE_NAMES[cfg]}, split-K {sk})"
    return f"dd3d::conv_igemm_f32[_dma]_kernel ({hip.TILE_NAMES[cfg]}, split-K {sk})"


def row_rings_default(np_, bm, bn, nwaves):
    """(NSB, NSA) of csrc/conv_planes_row.hip::RowRings for the product build (no -DDD3D_ROW_* knob); a CPU test compares it with
    dd3d_conv_row_rings for every tile and mode."""
    ast, bst = np_ * (bm + 16) * 64, np_ * bn * 64
    budget = (152 if (nwaves == 8 or 2 * ast > 65536) else 76) * 1024
    nsb = 3 if 2 * ast + 3 * bst <= budget else 2
    return nsb, 2


def tile_key(m_list, N, Kpad, stride):
    return f"{'+'.join(str(m) for m in m_list)},{N},{Kpad},{stride}"


def _load_tile_table(math_name):
    """Measured exceptions to the analytic model below: {tile_key: [tile, splitk, best_us, model_us]}, produced on an MI355X by
    tests/gpu_tile_explore.py (every candidate timed; entries kept only where the best beats the model's pick by > 5 %)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "data", f"tile_table_{math_name}.json")
    return json.load(open(path)) if os.path.exists(path) else {}


TILE_TABLE = {hip.MATH_F32: _load_tile_table("f32"), hip.MATH_BF16X3: _load_tile_table("bf16x3"), hip.MATH_BF16X2: _load_tile_table("bf16x2"),
              hip.MATH_BF16: _load_tile_table("bf16"), hip.MATH_F16X2: {}}
PLANE_TILE_TABLE = {m: _load_tile_table(n + "_planes") for n, m in (("bf16x3", hip.MATH_BF16X3), ("bf16x2", hip.MATH_BF16X2), ("bf16", hip.MATH_BF16),
                                                                     ("f16x2", hip.MATH_F16X2))}


def _tile_overrides():
    """DD3D_TILE_OVERRIDE: `key=tile:splitk` pairs separated by ';' (key as `tile_key` prints it) that win over the measured table --
    for sweeps of the issue mode, where the tile that minimises one launch's latency need not maximise the throughput of several slots."""
    spec = os.environ.get("DD3D_TILE_OVERRIDE", "")
    out = {}
    for item in filter(None, (s.strip() for s in spec.split(";"))):
        key, val = item.split("=")
        tile, _, sk = val.partition(":")
        out[key.strip()] = [tile.strip(), int(sk or 1)]
    return out


# Throughput policy (round 6): launch plans that run with OTHER plans in flight (PipelinedForward slots covering several requests) pay for
# CU-time, not for one launch's latency.  The short-K 3 x 3 convolutions of the backbone are bound by the LDS-DMA instructions a CU can issue
# (one 1-KiB piece per ~57 cycles), and most of a small tile's pieces are FILTER rows every block streams again: a 256-row tile with little
# split-K moves half the filter bytes per MFMA of the 128-row tile the latency table picks -- slower alone (fewer blocks: +8 % on a slot run
# by itself), faster in the issue mode (+2 % on the driver command, profiles/r06j_bm256_sweep.txt).  {tile_key: [tile, splitk]}, measured in situ.
THROUGHPUT_TILE_TABLE = {m: _load_tile_table(n + "_planes_throughput") for n, m in (("f16x2", hip.MATH_F16X2), )}


def default_tile_policy():
    """DD3D_TILE_POLICY=latency|throughput overrides what the runners choose (latency: one plan at a time; throughput: several in flight)."""
    v = os.environ.get("DD3D_TILE_POLICY", "").strip().lower()
    return v if v in ("latency", "throughput") else None


def preferred_tile(m_list, N, Kpad, stride=1, math=0, planes=False, policy="latency"):
    """(tile_cfg, splitk) that DD3D_TILE_OVERRIDE or -- for the throughput policy -- THROUGHPUT_TILE_TABLE names for this exact shape, else None:
    what wins even over a tile the plan builder asks for explicitly (the narrow predictors)."""
    key = tile_key(m_list, N, Kpad, stride)
    hit = _tile_overrides().get(key)
    if hit is None and planes and policy == "throughput":
        hit = THROUGHPUT_TILE_TABLE.get(math, {}).get(key)
    if hit is None:
        return None
    cfg = next(c for c, nm in hip.TILE_NAMES.items() if nm == hit[0])
    return (PLANE_TILE_ALIAS.get(cfg, cfg) if planes else cfg), int(hit[1])


def choose_tiling(m_list, N, Kpad, stride=1, math=0, planes=False, policy="latency"):
    """Pick (tile_cfg, splitk): a measured table entry when this exact shape has one, else minimise the modelled makespan
    on 256 CUs: every block costs BM*BN*K MACs on its CU's matrix pipe (partial tiles cost the same as full ones); split-K
    adds the partial-sum exchange.  `planes`: the split-plane-input kernel (its own measured table; the f32-input kernel's
    entries serve as the fallback for the three-term mode, their two-K-tiles-per-barrier variants mapped to the plain tile).
    `policy` "throughput": entries of THROUGHPUT_TILE_TABLE win over the latency table (plans that share the chip with other plans)."""
    allowed = PLANE_TILES if planes else MATH_TILES[math]
    # (measured and dropped: forcing the small convolutions onto 4-wave blocks with <= 48 KiB of LDS so that other streams' blocks could
    # share their CUs -- pipelined throughput 960 -> 921 img/s, profiles/r02_notes.md)
    key = tile_key(m_list, N, Kpad, stride)
    hit = PLANE_TILE_TABLE[math].get(key) if planes else TILE_TABLE[math].get(key)
    if planes and policy == "throughput":
        hit = THROUGHPUT_TILE_TABLE.get(math, {}).get(key, hit)
    forced = _tile_overrides().get(key)  # DD3D_TILE_OVERRIDE="M[+M..],N,Kpad,stride=tile:splitk;..." (measurement sweeps, tests/gpu_issue_sweep.sh)
    if forced is not None:
        hit = forced
    if hit is None and planes and math == hip.MATH_BF16X3:
        hit = TILE_TABLE[math].get(key)
    if hit is not None:
        cfg = next(c for c, nm in hip.TILE_NAMES.items() if nm == hit[0])
        return (PLANE_TILE_ALIAS.get(cfg, cfg) if planes else cfg), int(hit[1])
    nk = Kpad // 32
    # Large launches of the two-term modes with N >= 256: the 8-wave 256 x 256 tile (wave tile 128 x 64: one ds_read_b128 per two MFMAs) beat
    # every other tile by 7-9 % on every such shape measured in round 4 -- head towers, the merged FPN output launch, V2-99's 1 x 1 concat
    # convolutions, from 158 blocks (profiles/r04h_*, r04t_*) -- so it is the default there, not only where a table entry names it.
    if planes and hip.MATH_PLANES[math] <= 2 and N >= 256 and hip.TILE_256x256_W8 in allowed and -(-N // 256) * 256 <= 1.15 * N:
        if sum(-(-m // 256) for m in m_list) * -(-N // 256) >= 150:
            return hip.TILE_256x256_W8, 1
    best = None
    for cfg in allowed:
        if cfg in BIG_WAVE_TILES or cfg == hip.TILE_128x32_W4:
            continue  # (no analytic model: the measured table or an explicit `tile=` selects them)
        bm, bn = hip.TILE_SHAPES[cfg]
        if bn == 32 and N > 32:
            continue
        if bn > 32 and N <= 32 and not (planes and bn == 64):  # (the split-plane kernel has no 32-wide tile: narrow convs pad to 64)
            continue
        if bn == 128 and N <= 64:
            continue
        blocks = sum(-(-m // bm) for m in m_list) * -(-N // bn)
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (nk // sk < 4 or blocks >= NUM_CU):
                continue
            per = -(-nk // sk)
            cost = -(-blocks * sk // NUM_CU) * bm * bn * per * 32
            # measured matrix-pipe efficiency of each tile shape once the chip is full (profiles/r01b_conv_ops.txt):
            # 128x128 ~119 TF/s, 128x64 ~95, 64x64 ~74, 128x32 (N <= 32 pads the 32-wide MFMA) ~45
            if math == hip.MATH_F32:
                cost /= {(128, 128): 1.0, (128, 64): 0.80, (64, 128): 0.80, (64, 64): 0.63, (128, 32): 0.40}[(bm, bn)]
            else:  # split-bf16 kernel: ~2x the f32 rate on the big tiles, LDS-read bound on the small ones
                cost /= {hip.TILE_256x128: 2.4, hip.TILE_128x128: 1.8, hip.TILE_128x64: 1.3, hip.TILE_64x128: 1.3,
                         hip.TILE_128x128_W4: 1.0, hip.TILE_64x64_W4: 0.6, hip.TILE_128x64_W4: 0.8,
                         hip.TILE_128x64_K2: 1.0, hip.TILE_64x128_K2: 1.0, hip.TILE_64x64_W4K2: 0.5}[cfg]  # rough; the table decides
                if math in (hip.MATH_BF16X2, hip.MATH_BF16, hip.MATH_F16X2):  # fewer products per K-tile: the matrix term shrinks, the rest does not
                    cost *= {hip.MATH_BF16X2: 0.6, hip.MATH_F16X2: 0.6, hip.MATH_BF16: 0.35}[math]
            if sk > 1:
                # second launch (~2 us) + partial-sum round trip (sk*M*N*8 B at ~3 TB/s), in per-CU MAC units
                # (one CU retires 157.3e12 / 2 / 256 = 3.07e11 MAC/s)
                cost += 0.6e6 + sk * sum(m_list) * N * 8 / 3e12 * 3.07e11
            if best is None or cost < best[0]:
                best = (cost, cfg, sk)
    return best[1], best[2]

