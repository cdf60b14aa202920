"""ctypes loader for libbposd_mi355x.so (the C-ABI in include/bposd_mi355x.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``bp_osd_amd.build.build_library()`` with hipcc for gfx950.  There is no fallback:
if the library is missing, importing the decoder raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPOSD_LIB") or os.path.join(_HERE, "libbposd_mi355x.so")

BPOSD_OK = 0
BPOSD_ERR_INVALID = -1
BPOSD_ERR_UNSUPPORTED = -2
BPOSD_ERR_HIP = -3
BPOSD_ERR_NO_DEVICE = -4

# every symbol include/bposd_mi355x.h declares
EXPORTED_SYMBOLS = (
    "bposd_device_count",
    "bposd_version",
    "bposd_create",
    "bposd_update_channel_probs",
    "bposd_decode_batch",
    "bposd_decode_batch_packed",
    "bposd_decode_batch_async",
    "bposd_decode_batch_packed_async",
    "bposd_decode_batch_device",
    "bposd_decode_batch_device_packed",
    "bposd_decode_batch_select",
    "bposd_decode_batch_select_device",
    "bposd_pack_rows_device",
    "bposd_pack_rows_device_lane",
    "bposd_synchronize",
    "bposd_num_lanes",
    "bposd_last_lane",
    "bposd_synchronize_lane",
    "bposd_lane_timing",
    "bposd_host_alloc",
    "bposd_host_free",
    "bposd_last_timing",
    "bposd_info",
    "bposd_posterior_llr",
    "bposd_set_bp_variant",
    "bposd_set_osd_variant",
    "bposd_last_osd_kernel",
    "bposd_last_error",
    "bposd_destroy",
)

# include/bposd_mi355x_debug.h: diagnostics (which kernel ran, LDS layout models and tables)
DEBUG_SYMBOLS = (
    "bposd_layout_info",
    "bposd_bp_kernel_info",
    "bposd_debug_local_layout",
    "bposd_debug_class_layout",
)


class BposdConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("bp_method", C.c_int32),
        ("ms_scaling_factor", C.c_double),
        ("max_iter", C.c_int32),
        ("osd_method", C.c_int32),
        ("osd_order", C.c_int32),
        ("sort_tie_policy", C.c_int32),
        ("weight_fn", C.c_int32),
        ("schedule", C.c_int32),
        ("ps_clip", C.c_double),
        ("osd_e_bit_order", C.c_int32),
        ("ps_math_form", C.c_int32),
    ]


_lib = None


def load():
    """Load the HIP library; raises RuntimeError (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  bp_osd_amd has no CPU fallback."
        )
    # A handle's lanes are HIP streams that should run side by side; the runtime multiplexes a process's streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one take turns.  Read when the runtime starts, so this
    # only helps a process that has not touched the GPU yet; never overrides the caller's own setting.  (Measured, DESIGN.md
    # section 1: one synchronous host-to-host call of 131072 syndromes 30.7 -> 27.5 ms, BASELINE configs[4] as the fifth handle of
    # a process 10.0-10.4 k -> 12.1 k syndromes/s.)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.bposd_device_count.restype = C.c_int
    lib.bposd_version.restype = C.c_char_p
    lib.bposd_create.argtypes = [C.POINTER(BposdConfig), vp, vp, C.c_int32, C.c_int32, vp, C.POINTER(vp)]
    lib.bposd_create.restype = C.c_int
    lib.bposd_update_channel_probs.argtypes = [vp, vp]
    lib.bposd_update_channel_probs.restype = C.c_int
    lib.bposd_decode_batch.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch.restype = C.c_int
    lib.bposd_decode_batch_packed.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_packed.restype = C.c_int
    lib.bposd_decode_batch_async.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_async.restype = C.c_int
    lib.bposd_decode_batch_packed_async.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_packed_async.restype = C.c_int
    lib.bposd_decode_batch_device.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_device.restype = C.c_int
    lib.bposd_decode_batch_device_packed.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_device_packed.restype = C.c_int
    lib.bposd_decode_batch_select.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_select.restype = C.c_int
    lib.bposd_decode_batch_select_device.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.bposd_decode_batch_select_device.restype = C.c_int
    lib.bposd_pack_rows_device.argtypes = [vp, vp, C.c_int64, C.c_int32, vp]
    lib.bposd_pack_rows_device.restype = C.c_int
    lib.bposd_synchronize.argtypes = [vp]
    lib.bposd_synchronize.restype = C.c_int
    lib.bposd_num_lanes.argtypes = [vp]
    lib.bposd_num_lanes.restype = C.c_int
    lib.bposd_last_lane.argtypes = [vp]
    lib.bposd_last_lane.restype = C.c_int
    lib.bposd_synchronize_lane.argtypes = [vp, C.c_int32]
    lib.bposd_synchronize_lane.restype = C.c_int
    lib.bposd_lane_timing.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.bposd_lane_timing.restype = C.c_int
    lib.bposd_host_alloc.argtypes = [C.c_size_t]
    lib.bposd_host_alloc.restype = vp
    lib.bposd_host_free.argtypes = [vp]
    lib.bposd_host_free.restype = None
    lib.bposd_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.bposd_last_timing.restype = C.c_int
    lib.bposd_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int32)]
    lib.bposd_info.restype = C.c_int
    lib.bposd_layout_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.bposd_layout_info.restype = C.c_int
    lib.bposd_pack_rows_device_lane.argtypes = [vp, C.c_int32, vp, C.c_int64, C.c_int32, vp]
    lib.bposd_pack_rows_device_lane.restype = C.c_int
    lib.bposd_posterior_llr.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp]
    lib.bposd_posterior_llr.restype = C.c_int
    lib.bposd_bp_kernel_info.argtypes = [vp, C.POINTER(C.c_int32), vp]
    lib.bposd_bp_kernel_info.restype = C.c_int
    lib.bposd_debug_local_layout.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    lib.bposd_debug_local_layout.restype = C.c_int
    lib.bposd_debug_class_layout.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.bposd_debug_class_layout.restype = C.c_int
    lib.bposd_set_osd_variant.argtypes = [vp, C.c_int32]
    lib.bposd_set_osd_variant.restype = C.c_int
    lib.bposd_last_osd_kernel.argtypes = [vp]
    lib.bposd_last_osd_kernel.restype = C.c_int
    lib.bposd_set_bp_variant.argtypes = [vp, C.c_int32]
    lib.bposd_set_bp_variant.restype = C.c_int
    lib.bposd_last_error.argtypes = [vp]
    lib.bposd_last_error.restype = C.c_char_p
    lib.bposd_destroy.argtypes = [vp]
    lib.bposd_destroy.restype = None
    _lib = lib
    return lib


def check(lib, handle, rc):
    """Map a C-ABI return code to the exception the reference's users would expect."""
    if rc == BPOSD_OK:
        return
    msg = lib.bposd_last_error(handle)
    msg = msg.decode() if msg else f"error {rc}"
    if rc in (BPOSD_ERR_INVALID, BPOSD_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)
