"""ctypes view of the thin C-ABI HIP layer (include/clo_hip.h).

Loads cl_ops_amd/lib/libcl_ops_hip.so — the only implementation there is. If
the library is missing or does not load, importing this module raises: there
is no CPU or PyTorch fallback for the sort/scan path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLO_HIP_LIBRARY: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("CLO_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libcl_ops_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cl_ops_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C cl_ops_amd/csrc` (needs hipcc). There is no fallback path." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int


class DeviceProps(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("compute_units", ci), ("max_threads_per_block", ci),
                ("wavefront_size", ci), ("lds_bytes_per_block", sz), ("global_mem_bytes", sz),
                ("gcn_arch", C.c_char * 64)]


def _sig(name, restype, *argtypes):
    try:
        f = getattr(lib, name)
    except AttributeError:
        # The shipped library exports everything this file names (tests/test_boundary_cpu.py checks it): a missing symbol is an
        # error. Only an OLDER build loaded through CLO_HIP_LIBRARY for an A/B run may lack entry points added since.
        if os.environ.get("CLO_HIP_LIBRARY"):
            return None
        raise
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("clo_hip_device_count", ci, C.POINTER(ci))
_sig("clo_hip_set_device", ci, ci)
_sig("clo_hip_get_device", ci, C.POINTER(ci))
_sig("clo_hip_get_device_props", ci, ci, C.POINTER(DeviceProps))
_sig("clo_hip_stream_create", ci, C.POINTER(vp))
_sig("clo_hip_stream_create_high_priority", ci, C.POINTER(vp))
_sig("clo_hip_stream_destroy", ci, vp)
_sig("clo_hip_stream_synchronize", ci, vp)
_sig("clo_hip_malloc", ci, C.POINTER(vp), sz)
_sig("clo_hip_free", ci, vp)
_sig("clo_hip_memcpy_h2d_async", ci, vp, vp, sz, vp)
_sig("clo_hip_memcpy_d2h_async", ci, vp, vp, sz, vp)
_sig("clo_hip_memcpy_d2d_async", ci, vp, vp, sz, vp)
_sig("clo_hip_memset_async", ci, vp, ci, sz, vp)
_sig("clo_hip_host_register", ci, vp, sz)
_sig("clo_hip_host_unregister", ci, vp)
_sig("clo_hip_stream_is_capturing", ci, vp)
_sig("clo_hip_graph_capture_begin", ci, vp)
_sig("clo_hip_graph_capture_end", ci, vp, C.POINTER(vp))
_sig("clo_hip_graph_launch", ci, vp, vp)
_sig("clo_hip_graph_destroy", ci, vp)
_sig("clo_hip_timing_enabled", ci)
_sig("clo_hip_event_create", ci, C.POINTER(vp))
_sig("clo_hip_event_destroy", ci, vp)
_sig("clo_hip_event_record", ci, vp, vp)
_sig("clo_hip_event_synchronize", ci, vp)
_sig("clo_hip_event_query", ci, vp)
_sig("clo_hip_env_refresh", None)
_sig("clo_hip_event_elapsed_ms", ci, vp, vp, C.POINTER(C.c_float))
_sig("clo_hip_stream_wait_event", ci, vp, vp)
_sig("clo_hip_error_string", C.c_char_p, ci)
_sig("clo_hip_scan_workspace_bytes", sz, sz, ci, ci)
_sig("clo_hip_scan_workspace_init", ci, vp, sz, vp)
_sig("clo_hip_scan_workspace_forget", ci, vp)
_sig("clo_hip_scan_workspace_set_epoch", ci, vp, C.c_uint, vp)
_sig("clo_hip_scan_exclusive", ci, vp, vp, sz, ci, ci, ci, vp, sz, vp)
_sig("clo_hip_scan_exclusive_carry", ci, vp, vp, sz, ci, ci, ci, vp, vp, vp, sz, vp)
_sig("clo_hip_scan_fp_workspace_bytes", sz, sz, ci)
_sig("clo_hip_scan_is_typed", ci, ci, ci)
_sig("clo_hip_scan_typed_workspace_bytes", sz, sz, ci)
_sig("clo_hip_scan_exclusive_typed", ci, vp, vp, sz, ci, ci, vp, sz, vp)
_sig("clo_hip_scan_exclusive_fp", ci, vp, vp, sz, ci, ci, vp, sz, vp)
_sig("clo_hip_reduce_sum", ci, vp, sz, ci, ci, vp, vp)
_sig("clo_hip_radix_workspace_bytes", sz, sz, ci, ci, ci)
_sig("clo_hip_radix_polls", ci, sz, ci, ci)
_sig("clo_hip_radix_sort", ci, vp, vp, vp, sz, ci, ci, ci, ci, ci, vp, sz, vp)
_sig("clo_hip_radix_takes_first_digits", ci, sz, ci, ci, ci)
_sig("clo_hip_radix_sort_fed", ci, vp, vp, vp, sz, ci, ci, ci, ci, ci, vp, vp, sz, vp)
_sig("clo_hip_radix_seg_workspace_bytes", sz, sz, ci, ci, ci)
_sig("clo_hip_radix_sort_segmented", ci, vp, vp, vp, sz, C.POINTER(sz), ci, C.POINTER(sz), C.POINTER(sz), C.POINTER(ci), ci,
     ci, ci, ci, ci, vp, sz, vp, C.POINTER(ci))
_sig("clo_hip_radix_sort_segmented2", ci, vp, vp, vp, vp, sz, C.POINTER(sz), ci, C.POINTER(sz), C.POINTER(sz), C.POINTER(ci), C.POINTER(ci), ci,
     ci, ci, ci, ci, vp, sz, vp, C.POINTER(ci))
_sig("clo_hip_msd_histogram", ci, vp, sz, ci, ci, ci, ci, vp, vp)
_sig("clo_hip_msd_partition", ci, vp, vp, sz, ci, ci, ci, ci, vp, vp, sz, vp)
_sig("clo_hip_msd_workspace_bytes", sz, sz, ci, ci)
_sig("clo_hip_bitonic_padded_numel", sz, sz)
_sig("clo_hip_bitonic_simple", ci, vp, sz, ci, ci, ci, ci, ci, ci, C.POINTER(ci), vp)
_sig("clo_hip_bitonic_tiled", ci, vp, sz, ci, ci, ci, ci, ci, ci, C.POINTER(ci), vp)
_sig("clo_hip_bitonic_any", ci, vp, sz, ci, ci, ci, ci, ci, ci, C.POINTER(ci), vp)
_sig("clo_hip_kernel_lds_bytes", sz, C.c_char_p, ci, ci)
_sig("clo_hip_bitonic_jit_create", ci, ci, ci, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_char_p))
_sig("clo_hip_bitonic_jit_destroy", None, vp)
_sig("clo_hip_bitonic_jit_sort", ci, vp, vp, sz, ci, C.POINTER(ci), vp)
_sig("clo_hip_bitonic_jit_gselect", ci, vp, vp, vp, sz, vp)
_sig("clo_hip_check_status", ci, vp, vp)
_sig("clo_hip_rccl_unique_id", ci, vp)
_sig("clo_hip_rccl_comm_create", ci, C.POINTER(vp), vp, ci, ci)
_sig("clo_hip_rccl_comm_destroy", ci, vp)
_sig("clo_hip_rccl_comm_abort", ci, vp)
_sig("clo_hip_rccl_all_gather_u64", ci, vp, vp, vp, sz, vp)
_sig("clo_hip_rccl_all_to_all_v", ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("clo_hip_set_launch_observer", ci, vp, vp)
_sig("clo_hip_radix_jit_create", ci, ci, ci, C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_char_p))
_sig("clo_hip_radix_jit_destroy", None, vp)
_sig("clo_hip_radix_jit_sort", ci, vp, vp, vp, vp, vp, sz, ci, vp, sz, vp)
_sig("clo_hip_timing_enable", ci, ci)
_sig("clo_hip_timing_reset", ci)
_sig("clo_hip_timing_read", ci, C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_float))


CLO_HIP_EARGS, CLO_HIP_EUNSUPPORTED, CLO_HIP_EWORKSPACE, CLO_HIP_ETIMEOUT = -1, -2, -3, -4   # include/clo_hip.h


class HipError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib.clo_hip_error_string(status)
        super().__init__("%s: %s (status %d)" % (what or "clo_hip", msg.decode() if msg else "?", status))


def check(status, what=""):
    if status != 0:
        raise HipError(status, what)


def device_count():
    n = ci(0)
    st = lib.clo_hip_device_count(C.byref(n))
    return n.value if st == 0 else 0


def timing_read(label):
    """(launch count, total ms) recorded under `label` since the last reset."""
    n, ms = C.c_uint(0), C.c_float(0)
    check(lib.clo_hip_timing_read(label.encode(), C.byref(n), C.byref(ms)), "clo_hip_timing_read")
    return n.value, ms.value
