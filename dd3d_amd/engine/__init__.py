"""Launch-plan builder and executor of the DD3D forward path on MI355X (one package, split by concern in round 5):

  packing    activation buffers (f32 NHWC / split planes), filter packing and the 16-bit term splits
  tiling     tile + split-K choice (measured tables, analytic fallback), kernel names as rocprofv3 prints them
  ops        one object per libdd3d_hip launch (ConvOp, FusedStemOp, SmallcConvOp, CallOp)
  plan       PlanBase: buffers, the model's weight store, op helpers, launch / hipGraph capture / replay, the f16x2 range guard
  backbones  DLA / VoVNet-V2 / FPN lowering (mixin)
  forward    ForwardPlan (trunk, heads, select / decode / NMS, the exchange record) and DenseDepthPlan

Everything is re-exported here: `from dd3d_amd.engine import ForwardPlan, ConvOp, choose_tiling, ...` keeps working.
"""
from dd3d_amd.engine.packing import Buf, View, dense_filter, pack_filter, pack_smallc_bf16x3, pack_smallc_f16x2, pad32, scatter_in_channels, split_bf16x3, split_f16x2_host, split_planes_host  # noqa: F401
from dd3d_amd.engine.tiling import (BIG_WAVE_TILES, BLOCKS_PER_CU, MATH_NAMES, MATH_TILES, NUM_CU, PLANE_TILE_ALIAS, PLANE_TILE_TABLE, PLANE_TILES, TILE_TABLE,  # noqa: F401
                                    TILE_WAVE_GRID, _tile_overrides, choose_tiling, default_math, kernel_signature, row_rings_default, tile_key)
from dd3d_amd.engine.ops import CallOp, ConvOp, FusedStemOp, OpList, SmallcConvOp  # noqa: F401
from dd3d_amd.engine.plan import HalfRangeOverflow, HalfRangeUnderflow, PlanBase, relax_arithmetic  # noqa: F401
from dd3d_amd.engine.backbones import BackboneLowering  # noqa: F401
from dd3d_amd.engine.forward import DenseDepthPlan, ForwardPlan  # noqa: F401
