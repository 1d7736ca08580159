"""Dev tool (GPU): bench.py with every compute stream of the issue mode confined to its own set of CUs (hipExtStreamCreateWithCUMask).

The what-if sweep (profiles/r05n_*) says the tower launches and the rest of the forwards do not overlap: a tower launch owns every CU.  Question:
do five slots on five DISJOINT CU partitions (each slot's launches packed 5x denser on a fifth of the chip: more waves per SIMD for the
latency-bound small convolutions, no interference between slots) beat five slots time-sharing the whole chip?

    DD3D_CU_PARTS=5 python tests/gpu_cumask_bench.py --steps 20 --warmup 5     (DD3D_CU_PARTS=0: no masks = bench.py)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from dd3d_amd import parallel as P  # noqa: E402

PARTS = int(os.environ.get("DD3D_CU_PARTS", "5"))
NUM_CU = 256
_hip = C.CDLL("libamdhip64.so")
_hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
_made = []


def masked_stream(part):
    lo, hi = part * NUM_CU // PARTS, (part + 1) * NUM_CU // PARTS
    words = (C.c_uint32 * (NUM_CU // 32))()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(C.byref(st), NUM_CU // 32, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed ({rc})"
    _made.append(st)
    return torch.cuda.ExternalStream(st.value)


_orig_stream = P._DeviceRuntime.stream
_count = [0]


def stream(self):
    # PipelinedForward creates its compute streams first, then the post stream (unmasked: the NMS tail may run anywhere)
    i = _count[0]
    _count[0] += 1
    want = int(os.environ.get("DD3D_BENCH_COMPUTE_STREAMS", "5"))
    if PARTS > 0 and i < want:
        return masked_stream(i % PARTS)
    return _orig_stream(self)


P._DeviceRuntime.stream = stream

if __name__ == "__main__":
    sys.argv += ["--no-cpu-baseline"]
    bench.main()
