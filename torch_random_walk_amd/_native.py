"""Loads the native pieces; fails loudly when they are missing."""
import ctypes
import os

import torch  # noqa: F401  first: the extension must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libtrw_hip.so")

if not os.path.exists(HIP_LIB_PATH):
    raise ImportError(
        "torch_random_walk_amd: %s is missing. Build it with `python torch_random_walk_amd/_build.py` "
        "(hipcc --offload-arch=gfx950). Nothing stands in for it: GPU kernels and host legs both live there." % HIP_LIB_PATH)

try:
    from . import torch_rw_native as ext  # the pybind module, same name as the reference's
except ImportError as e:  # pragma: no cover - exercised only on a broken build
    raise ImportError(
        "torch_random_walk_amd: the torch_rw_native extension failed to load (%s). Rebuild with "
        "`python torch_random_walk_amd/_build.py --force`. Nothing stands in for it." % e) from e

RNG_AUTO = int(ext.RNG_AUTO)
RNG_GLIBC = int(ext.RNG_GLIBC)
RNG_WALKER = int(ext.RNG_WALKER)

_cabi = None


def cabi():
    """ctypes view of the C ABI declared in include/trw.h (same .so the extension links)."""
    global _cabi
    if _cabi is None:
        L = ctypes.CDLL(HIP_LIB_PATH)
        vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
        L.trw_version.restype = ctypes.c_int
        L.trw_last_error.restype = ctypes.c_char_p
        L.trw_walk_csr_workspace_bytes.argtypes = [i64, i64, i64, i32, dbl, dbl, i32]
        L.trw_walk_csr_workspace_bytes.restype = ctypes.c_size_t
        L.trw_walk_csr_hip.argtypes = [vp, i64, vp, i64, vp, i64, i64, dbl, dbl, i32, i32, i32, vp, vp,
                                       ctypes.c_size_t, vp]
        L.trw_walk_csr_hip.restype = ctypes.c_int
        L.trw_walk_triples_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i32]
        L.trw_walk_triples_workspace_bytes.restype = ctypes.c_size_t
        L.trw_walk_triples_hip.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i64, i32, i32, vp, vp,
                                           ctypes.c_size_t, vp]
        L.trw_walk_triples_hip.restype = ctypes.c_int
        L.trw_walk_edge_list_workspace_bytes.argtypes = [i64, i64, i64, i32, i64, i32]
        L.trw_walk_edge_list_workspace_bytes.restype = ctypes.c_size_t
        L.trw_walk_edge_list_hip.argtypes = [vp, i64, vp, i64, vp, i64, i64, dbl, dbl, i32, i32, i64, i32, i32, vp,
                                             vp, ctypes.c_size_t, vp]
        L.trw_walk_edge_list_hip.restype = ctypes.c_int
        L.trw_prepare_plan.argtypes = [i64, i64, i64, i32, dbl, dbl, i32]
        L.trw_prepare_plan.restype = ctypes.c_int32
        L.trw_prepare_csr_bytes.argtypes = [i64, i64, i32]
        L.trw_prepare_csr_bytes.restype = ctypes.c_size_t
        L.trw_prepare_csr_hip.argtypes = [vp, i64, vp, i64, i32, vp, ctypes.c_size_t, vp]
        L.trw_prepare_csr_hip.restype = ctypes.c_int
        L.trw_prepare_csr_persistent_bytes.argtypes = [i64, i64, i32]
        L.trw_prepare_csr_persistent_bytes.restype = ctypes.c_size_t
        L.trw_prepare_csr_scratch_bytes.argtypes = [i64, i64, i32]
        L.trw_prepare_csr_scratch_bytes.restype = ctypes.c_size_t
        L.trw_prepare_csr_scratch_hip.argtypes = [vp, i64, vp, i64, i32, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]
        L.trw_prepare_csr_scratch_hip.restype = ctypes.c_int
        L.trw_prepare_csr_region.argtypes = [i64, i64, i32, i32, ctypes.POINTER(ctypes.c_size_t),
                                             ctypes.POINTER(ctypes.c_size_t)]
        L.trw_prepare_csr_region.restype = ctypes.c_int
        L.trw_walk_csr_prepared_workspace_bytes.argtypes = [i64, i32, dbl, dbl, i32]
        L.trw_walk_csr_prepared_workspace_bytes.restype = ctypes.c_size_t
        L.trw_walk_csr_prepared_hip.argtypes = [vp, ctypes.c_size_t, i32, vp, i64, vp, i64, vp, i64, i64, dbl, dbl,
                                                i32, i32, i32, vp, vp, ctypes.c_size_t, vp]
        L.trw_walk_csr_prepared_hip.restype = ctypes.c_int
        L.trw_walk_csr_cpu.argtypes = [vp, i64, vp, i64, vp, i64, i64, dbl, dbl, i32, i32, i32, vp, i32]
        L.trw_walk_csr_cpu.restype = ctypes.c_int
        L.trw_status_post.argtypes = [vp, vp]
        L.trw_status_post.restype = ctypes.c_int
        L.trw_status_collect.argtypes = [ctypes.c_int, ctypes.POINTER(i64)]
        L.trw_status_collect.restype = ctypes.c_int
        L.trw_prepare_edge_list_plan.argtypes = [i64, i64, i64, i32, dbl, dbl, i64, i32]
        L.trw_prepare_edge_list_plan.restype = ctypes.c_int32
        L.trw_prepare_edge_list_bytes.argtypes = [i64, i64, i64, i32]
        L.trw_prepare_edge_list_bytes.restype = ctypes.c_size_t
        L.trw_prepare_edge_list_hip.argtypes = [vp, i64, vp, i64, i64, i32, vp, ctypes.c_size_t, vp]
        L.trw_prepare_edge_list_hip.restype = ctypes.c_int
        L.trw_walk_edge_list_prepared_hip.argtypes = [vp, ctypes.c_size_t, i32, vp, i64, vp, i64, vp, i64, i64, dbl, dbl,
                                                      i32, i32, i64, i32, i32, vp, vp, ctypes.c_size_t, vp]
        L.trw_walk_edge_list_prepared_hip.restype = ctypes.c_int
        L.trw_walk_status.argtypes = [vp, vp, ctypes.POINTER(i64)]
        L.trw_walk_status.restype = ctypes.c_int
        L.trw_accept_threshold.argtypes = [dbl]
        L.trw_accept_threshold.restype = ctypes.c_uint32
        L.trw_glibc_window_at.argtypes = [i32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
        L.trw_glibc_window_at.restype = None
        L.trw_profile_enable.argtypes = [ctypes.c_int]
        L.trw_profile_enable.restype = ctypes.c_int
        L.trw_profile_last.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3
        L.trw_profile_last.restype = ctypes.c_int
        L.trw_profile_prepare_last.argtypes = [ctypes.POINTER(ctypes.c_double)]
        L.trw_profile_prepare_last.restype = ctypes.c_int
        L.trw_walk_counters.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint64)]
        L.trw_walk_counters.restype = ctypes.c_int
        L.trw_selftest_hip.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint64)]
        L.trw_selftest_hip.restype = ctypes.c_int
        _cabi = L
    return _cabi
