"""Python handle on the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/mm_oracle.c header).

Builds oracle/libmm_oracle.so on demand and wraps it in the same BatchedMergeEnv plumbing the
product uses, on host tensors.  Import this only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libmm_oracle.so")
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from marl_mass_amd import _cabi as abi  # noqa: E402
from marl_mass_amd.vec_env import BatchedMergeEnv  # noqa: E402

_LIB = None


def build(force=False):
    # make decides what is stale (mm_oracle.c and every header of include/ are its prerequisites); it is not run at all
    # only where the sources did not travel (a box that received the built library alone)
    if force or not os.path.exists(LIB) or os.path.exists(os.path.join(HERE, "mm_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "libmm_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB


def library():
    global _LIB
    if _LIB is None:
        _LIB = abi.CLib(build())
        lib = _LIB.lib
        d, i32 = C.c_double, C.c_int32
        lib.orc_closest_lane.argtypes = [d, d, d]
        lib.orc_next_lane.argtypes = [i32, d, d]
        lib.orc_lane_local.argtypes = [i32, d, d, C.POINTER(d), C.POINTER(d)]
        lib.orc_lane_local.restype = None
        lib.orc_lane_heading_at.argtypes = [i32, d]
        lib.orc_lane_heading_at.restype = d
        lib.orc_lane_distance_with_heading.argtypes = [i32, d, d, d]
        lib.orc_lane_distance_with_heading.restype = d
        for n in ("orc_on_lane", "orc_is_reachable_from", "orc_after_end"):
            getattr(lib, n).argtypes = [i32, d, d]
        lib.orc_steering_control.argtypes = [d, d, d, d, i32]
        lib.orc_steering_control.restype = d
        lib.orc_get_corner.argtypes = [d, d, d, i32, C.POINTER(d), C.POINTER(d)]
        lib.orc_get_corner.restype = None
        lib.orc_speed_to_index.argtypes = [d]
        lib.orc_wrap_to_pi.argtypes = [d]
        lib.orc_wrap_to_pi.restype = d
        lib.orc_rect_intersect.argtypes = [d] * 10
        vp = C.c_void_p
        lib.orc_set_threads.argtypes = [i32]
        lib.orc_set_math.argtypes = [i32]
        lib.orc_batch_pose.argtypes = [i32] + [vp] * 11
        lib.orc_batch_steering.argtypes = [i32] + [vp] * 7
        lib.orc_batch_rect.argtypes = [i32] + [vp] * 3
        for n in ("orc_batch_pose", "orc_batch_steering", "orc_batch_rect"):
            getattr(lib, n).restype = None
    return _LIB


def set_math_mode(mode):
    """0: platform libm (= the reference's arithmetic, golden-pinned); 1: include/mm_math.h (= the HIP path's)."""
    return library().lib.orc_set_math(int(mode))


def OracleEnv(E, N, env_id="merge-multi-agent-v1", config=None, **kw):
    """BatchedMergeEnv backed by the CPU oracle on host tensors."""
    kw.pop("device", None)
    return BatchedMergeEnv(library(), E, N, env_id=env_id, config=config, device="cpu", **kw)
