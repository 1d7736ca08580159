"""ctypes binding of libdd3d_hip.so (C ABI declared in include/dd3d_hip.h).

The library is the ONLY compute path of this package: importing ``lib()`` raises if the shared object
has not been built (``python -c "import __graft_entry__ as g; g.build()"``) -- there is no eager / CPU
fallback anywhere in ``dd3d_amd``.
"""
import ctypes as C
import os

import numpy as np

# DD3D_HIP_LIB: alternate build of the same ABI (A/B measurements of two kernel versions on one box, tests/tools/ab_lib.sh)
_LIB_PATH = os.environ.get("DD3D_HIP_LIB") or os.path.join(os.path.dirname(__file__), "lib", "libdd3d_hip.so")
_lib = None

MAX_LEVELS = 8
MATH_F32, MATH_BF16X3, MATH_BF16X2, MATH_BF16, MATH_F16X2 = 0, 1, 2, 3, 4
MATH_PLANES = {MATH_F32: 0, MATH_BF16X3: 3, MATH_BF16X2: 2, MATH_BF16: 1, MATH_F16X2: 2}  # 16-bit terms per value (dd3d_math_planes)
ABI_VERSION = 6
STATUS_F16_OVERFLOW = 1
STATUS_CHAIN_TIMEOUT = 2  # a block of a chain launch (dd3d_conv_launch.chain) gave up waiting for its producers
CAND_FIELDS = 22
DET_FIELDS = 32

TILE_128x128, TILE_128x64, TILE_64x64, TILE_128x32, TILE_64x128, TILE_256x128, TILE_128x128_W4, TILE_64x64_W4, TILE_128x64_W4, \
    TILE_128x64_K2, TILE_64x128_K2, TILE_64x64_W4K2, TILE_256x128_T42, TILE_128x256_T24, TILE_256x256_W8, TILE_128x32_W4, \
    TILE_192x256_W8 = range(17)
TILE_SHAPES = {TILE_128x128: (128, 128), TILE_128x64: (128, 64), TILE_64x64: (64, 64), TILE_128x32: (128, 32),
               TILE_64x128: (64, 128), TILE_256x128: (256, 128), TILE_128x128_W4: (128, 128), TILE_64x64_W4: (64, 64),
               TILE_128x64_W4: (128, 64), TILE_128x64_K2: (128, 64), TILE_64x128_K2: (64, 128), TILE_64x64_W4K2: (64, 64),
               TILE_256x128_T42: (256, 128), TILE_128x256_T24: (128, 256), TILE_256x256_W8: (256, 256), TILE_128x32_W4: (128, 32),
               TILE_192x256_W8: (192, 256)}
TILE_NAMES = {TILE_128x128: "128x128", TILE_128x64: "128x64", TILE_64x64: "64x64", TILE_128x32: "128x32", TILE_64x128: "64x128",
              TILE_256x128: "256x128", TILE_128x128_W4: "128x128w4", TILE_64x64_W4: "64x64w4", TILE_128x64_W4: "128x64w4", TILE_128x64_K2: "128x64k2",
              TILE_64x128_K2: "64x128k2", TILE_64x64_W4K2: "64x64w4k2", TILE_256x128_T42: "256x128t42", TILE_128x256_T24: "128x256t24",
              TILE_256x256_W8: "256x256w8", TILE_128x32_W4: "128x32w4", TILE_192x256_W8: "192x256w8"}

# numpy mirror of `dd3d_conv_seg` (120 bytes) -- arrays of it are uploaded to the device as raw bytes.
CONV_SEG_DTYPE = np.dtype(
    [("in_", "<u8"), ("w", "<u8"), ("scale", "<u8"), ("bias", "<u8"), ("lo", "<u8"), ("res", "<u8"), ("out", "<u8"),
     ("B", "<i4"), ("H", "<i4"), ("W", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"), ("in_pitch", "<i4"), ("out_pitch", "<i4"),
     ("res_pitch", "<i4"), ("M", "<i4"), ("res_mode", "<i4"), ("in_planes", "<u8"), ("n_limit", "<i4"),
     ("reserved", "<i4"), ("out_planes", "<u8")]
)
assert CONV_SEG_DTYPE.itemsize == 120


class ConvLaunch(C.Structure):
    """`dd3d_conv_launch`."""
    _fields_ = [
        ("segs", C.c_void_p), ("tiles", C.c_void_p), ("workspace", C.c_void_p), ("nsegs", C.c_int32), ("ntiles", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("Cin", C.c_int32), ("N", C.c_int32),
        ("Kpad", C.c_int32), ("Npad", C.c_int32), ("relu", C.c_int32), ("splitk", C.c_int32), ("math_mode", C.c_int32),
        ("tile_cfg", C.c_int32), ("zero_page", C.c_void_p), ("tile_counters", C.c_void_p), ("seg0_host", C.c_void_p), ("in_relu", C.c_int32),
        ("in_planes", C.c_int32), ("out_plane_scale", C.c_float), ("status", C.c_void_p), ("amax", C.c_void_p), ("chain", C.c_int32),
        ("chain_sync", C.c_void_p), ("chain_tile0", C.c_void_p)
    ]


class SmallcArgs(C.Structure):
    """`dd3d_smallc_args`."""
    _fields_ = [
        ("in_", C.c_void_p), ("out", C.c_void_p), ("w3", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p), ("lo", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("in_pitch", C.c_int32),
        ("out_pitch", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("N", C.c_int32), ("relu", C.c_int32)
    ]


class StemArgs(C.Structure):
    """`dd3d_stem_args`."""
    _fields_ = [
        ("src", C.c_void_p), ("sizes", C.c_void_p), ("mean", C.c_float * 3), ("stdv", C.c_float * 3), ("w1", C.c_void_p), ("scale1", C.c_void_p),
        ("bias1", C.c_void_p), ("w2", C.c_void_p), ("scale2", C.c_void_p), ("bias2", C.c_void_p), ("w3", C.c_void_p), ("scale3", C.c_void_p),
        ("bias3", C.c_void_p), ("out", C.c_void_p), ("out_planes", C.c_void_p), ("B", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32),
        ("out_pitch", C.c_int32), ("plane_scale", C.c_float), ("status", C.c_void_p), ("K", C.c_void_p), ("inv_K", C.c_void_p),
        ("zero_f32", C.c_void_p), ("zero_count", C.c_int32)
    ]


class ResizeArgs(C.Structure):
    """`dd3d_resize_args`."""
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p), ("tmp", C.c_void_p), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("new_h", C.c_int32), ("new_w", C.c_int32), ("src_plane", C.c_int64), ("dst_plane", C.c_int64), ("src_row", C.c_int32),
        ("dst_row", C.c_int32), ("lo_w", C.c_void_p), ("cnt_w", C.c_void_p), ("kk_w", C.c_void_p), ("lo_h", C.c_void_p),
        ("cnt_h", C.c_void_p), ("kk_h", C.c_void_p), ("ksize_w", C.c_int32), ("ksize_h", C.c_int32)
    ]


class SelectArgs(C.Structure):
    """`dd3d_select_args`."""
    _fields_ = [
        ("cls", C.c_void_p * MAX_LEVELS), ("box2d", C.c_void_p * MAX_LEVELS), ("box3d", C.c_void_p * MAX_LEVELS),
        ("H", C.c_int32 * MAX_LEVELS), ("W", C.c_int32 * MAX_LEVELS), ("stride", C.c_int32 * MAX_LEVELS),
        ("cls_pitch", C.c_int32), ("b2d_pitch", C.c_int32), ("b3d_pitch", C.c_int32), ("num_levels", C.c_int32),
        ("B", C.c_int32), ("num_classes", C.c_int32), ("class_agnostic_3d", C.c_int32), ("loc_offset_half", C.c_int32),
        ("thresh_with_ctr", C.c_int32), ("topk", C.c_int32), ("attr_off", C.c_int32), ("num_attr", C.c_int32),
        ("speed_off", C.c_int32), ("pre_nms_thresh", C.c_float), ("min_depth", C.c_float),
        ("max_depth", C.c_float), ("focal_factor", C.c_float), ("scale_depth_by_focal", C.c_int32), ("allocentric", C.c_int32),
        ("depth_is_distance", C.c_int32), ("inv_K", C.c_void_p), ("canon_sizes", C.c_void_p), ("scratch_idx", C.c_void_p),
        ("scratch_score", C.c_void_p), ("scratch_off", C.c_int64 * MAX_LEVELS), ("scratch_img_stride", C.c_int64),
        ("cand", C.c_void_p), ("counts", C.c_void_p), ("npass", C.c_void_p), ("slot_off", C.c_int32 * (MAX_LEVELS + 1))
    ]


class NmsArgs(C.Structure):
    """`dd3d_nms_args`."""
    _fields_ = [
        ("cand", C.c_void_p), ("counts", C.c_void_p), ("G", C.c_int32), ("num_levels", C.c_int32), ("topk", C.c_int32),
        ("do_nms", C.c_int32), ("use_score3d", C.c_int32), ("nms_thresh", C.c_float), ("post_topk", C.c_int32),
        ("do_postprocess", C.c_int32), ("out_size", C.c_void_p), ("sort_idx", C.c_void_p), ("sbox", C.c_void_p),
        ("scls", C.c_void_p), ("mask", C.c_void_p), ("nvalid", C.c_void_p), ("det", C.c_void_p), ("det_count", C.c_void_p),
        ("det_cap", C.c_int32), ("slot_off", C.c_int32 * (MAX_LEVELS + 1)), ("img_first", C.c_int32), ("img_per_rec", C.c_int32),
        ("rec_stride", C.c_int64)
    ]


class BevArgs(C.Structure):
    """`dd3d_bev_args`."""
    _fields_ = [
        ("det_in", C.c_void_p), ("count_in", C.c_void_p), ("inv_K", C.c_void_p), ("pose", C.c_void_p), ("group", C.c_void_p),
        ("out_size", C.c_void_p), ("G", C.c_int32), ("det_cap", C.c_int32), ("num_classes", C.c_int32),
        ("iou_thresh", C.c_float), ("max_dets", C.c_int32), ("write_global", C.c_int32),
        ("do_postprocess", C.c_int32), ("work", C.c_void_p), ("sbox", C.c_void_p), ("mask", C.c_void_p), ("meta", C.c_void_p),
        ("det_out", C.c_void_p), ("count_out", C.c_void_p), ("img_first", C.c_int32), ("img_per_rec", C.c_int32), ("rec_stride", C.c_int64)
    ]


EXPORTS = [
    "dd3d_abi_version", "dd3d_last_error", "dd3d_arch", "dd3d_build_flags", "dd3d_conv_tile_shape", "dd3d_conv_row_rings", "dd3d_conv2d_igemm_f32",
    "dd3d_preprocess_u8_nhwc4", "dd3d_maxpool2x2_nhwc", "dd3d_maxpool3x3s2_ceil_nhwc", "dd3d_ese_nhwc", "dd3d_upsample2x_add_nhwc", "dd3d_fcos_select_decode",
    "dd3d_invert_intrinsics", "dd3d_nms_finalize", "dd3d_bev_nms_aggregate", "dd3d_conv2d_smallc_supported", "dd3d_conv2d_smallc_bf16x3", "dd3d_rotate_iou_eval", "dd3d_d3_box_overlap", "dd3d_image_box_overlap", "dd3d_aligned_bilinear_scale", "dd3d_resize_bilinear_u8",
    "dd3d_format_boxes3d", "dd3d_math_planes", "dd3d_split_planes", "dd3d_maxpool2x2_planes", "dd3d_maxpool2x2_planes_in", "dd3d_upsample2x_add_planes", "dd3d_ese_fused", "dd3d_stem_fused_f16x2", "dd3d_fold_range_flags", "dd3d_pack_readback"
]


class HipLibraryMissing(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def build_flags():
    """Build-time knobs of the loaded library ("" = the product build)."""
    return lib().dd3d_build_flags().decode()


def lib():
    """Load (once) and return the ctypes handle.  Raises HipLibraryMissing if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "dd3d_amd has no CPU / eager fallback."
        )
    L = C.CDLL(_LIB_PATH)
    L.dd3d_abi_version.restype = C.c_int
    L.dd3d_last_error.restype = C.c_char_p
    L.dd3d_arch.restype = C.c_char_p
    if hasattr(L, "dd3d_build_flags"):
        L.dd3d_build_flags.restype = C.c_char_p
    L.dd3d_conv_tile_shape.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.dd3d_conv_row_rings.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.dd3d_conv2d_igemm_f32.argtypes = [C.POINTER(ConvLaunch), C.c_void_p]
    L.dd3d_preprocess_u8_nhwc4.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
        C.c_void_p
    ]
    L.dd3d_maxpool2x2_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    L.dd3d_upsample2x_add_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    L.dd3d_maxpool3x3s2_ceil_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    L.dd3d_ese_nhwc.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 7 + [C.c_void_p]
    L.dd3d_fcos_select_decode.argtypes = [C.POINTER(SelectArgs), C.c_void_p]
    L.dd3d_invert_intrinsics.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.dd3d_nms_finalize.argtypes = [C.POINTER(NmsArgs), C.c_void_p]
    L.dd3d_bev_nms_aggregate.argtypes = [C.POINTER(BevArgs), C.c_void_p]
    L.dd3d_conv2d_smallc_supported.argtypes = [C.c_int32] * 6
    L.dd3d_conv2d_smallc_bf16x3.argtypes = [C.POINTER(SmallcArgs), C.c_void_p]
    L.dd3d_stem_fused_f16x2.argtypes = [C.POINTER(StemArgs), C.c_void_p]
    L.dd3d_fold_range_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    L.dd3d_pack_readback.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.dd3d_rotate_iou_eval.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p]
    L.dd3d_d3_box_overlap.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p]
    L.dd3d_image_box_overlap.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p]
    L.dd3d_aligned_bilinear_scale.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 6 + [C.c_float, C.c_void_p]
    L.dd3d_resize_bilinear_u8.argtypes = [C.POINTER(ResizeArgs), C.c_void_p]
    L.dd3d_format_boxes3d.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p]
    L.dd3d_math_planes.argtypes = [C.c_int32]
    L.dd3d_maxpool2x2_planes.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 7 + [C.c_float, C.c_void_p, C.c_void_p]
    L.dd3d_maxpool2x2_planes_in.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]
    L.dd3d_upsample2x_add_planes.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 7 + [C.c_float, C.c_void_p, C.c_void_p]
    L.dd3d_ese_fused.argtypes = [C.c_void_p] * 9 + [C.c_int32] * 8 + [C.c_float, C.c_void_p, C.c_void_p]
    L.dd3d_split_planes.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_float, C.c_void_p, C.c_void_p]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError if the .so is stale
    assert L.dd3d_abi_version() == ABI_VERSION, "libdd3d_hip.so ABI version mismatch; rebuild"
    # A library compiled with build-time knobs (-DDD3D_...: ring depths, schedules, store forms -- A/B variants of tests/tools) says so;
    # it is only accepted when the caller chose it (DD3D_HIP_LIB names it, or DD3D_ALLOW_VARIANT_LIB=1), never as the default library.
    flags = L.dd3d_build_flags().decode()
    if flags and not (os.environ.get("DD3D_HIP_LIB") or os.environ.get("DD3D_ALLOW_VARIANT_LIB") == "1"):
        raise HipLibraryMissing(f"{_LIB_PATH} was built with non-default knobs ({flags}): rebuild it with __graft_entry__.build(force=True), "
                                "or select a variant explicitly with DD3D_HIP_LIB / DD3D_ALLOW_VARIANT_LIB=1")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libdd3d_hip {what} failed (rc={rc}): {lib().dd3d_last_error().decode()}")


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
