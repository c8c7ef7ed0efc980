"""Python view of the cl_ops sort/scan C API exported by libcl_ops_hip.so.

This is a thin ctypes wrapper — every call goes through the C-ABI
(include/clo_sort.h, clo_scan.h, clo_ccl.h), i.e. through the same entry
points a C program written against the reference's headers would use:
``clo_sort_new`` / ``clo_sort_with_device_data`` / ``clo_sort_with_host_data``
(reference: src/cl_ops/sort/clo_sort_abstract.in.h:116-170) and the
``clo_scan_*`` twins (src/cl_ops/scan/clo_scan_abstract.in.h:109-162).
There is no computation in Python and no fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _hip
from ._hip import lib, vp, sz, ci

# CloType numbering: src/cl_ops/common/clo_common.in.h:108-120
CLO_TYPES = {"char": 0, "uchar": 1, "short": 2, "ushort": 3, "int": 4, "uint": 5,
             "long": 6, "ulong": 7, "half": 8, "float": 9, "double": 10}
_NP_TO_CLO = {np.dtype(np.int8): "char", np.dtype(np.uint8): "uchar", np.dtype(np.int16): "short",
              np.dtype(np.uint16): "ushort", np.dtype(np.int32): "int", np.dtype(np.uint32): "uint",
              np.dtype(np.int64): "long", np.dtype(np.uint64): "ulong", np.dtype(np.float16): "half",
              np.dtype(np.float32): "float", np.dtype(np.float64): "double"}
CLO_TYPE_NP = {v: k for k, v in _NP_TO_CLO.items()}

CL_QUEUE_PROFILING_ENABLE = 1 << 1

# clo_error_codes: clo_common.in.h:80-95
CLO_SUCCESS, CLO_ERROR_ARGS, CLO_ERROR_IMPL_NOT_FOUND, CLO_ERROR_UNKNOWN_TYPE, CLO_ERROR_LIBRARY = 0, 2, 5, 6, 7


class GError(C.Structure):
    _fields_ = [("domain", C.c_uint32), ("code", C.c_int), ("message", C.c_char_p)]


GErrorP = C.POINTER(GError)
_u32 = C.c_uint32


def _sig(name, restype, *argtypes):
    try:
        f = getattr(lib, name)
    except AttributeError:
        if os.environ.get("CLO_HIP_LIBRARY"):   # (an older build in an A/B run: cl_ops_amd/_hip.py)
            return None
        raise
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_E = C.POINTER(GErrorP)
_sig("clo_gerror_free", None, GErrorP)
_sig("clo_quark_to_string", C.c_char_p, _u32)
_sig("clo_error_quark", _u32)
_sig("ccl_hip_error_quark", _u32)
_sig("clo_type_get_name", C.c_char_p, ci)
_sig("clo_type_sizeof", sz, ci)
_sig("clo_type_by_name", ci, C.c_char_p, _E)
_sig("clo_nlpo2", C.c_uint, C.c_uint)
_sig("clo_ones32", C.c_uint, C.c_uint)
_sig("clo_tzc", C.c_uint, ci)
_sig("clo_sum", C.c_uint, C.c_uint)

_sig("ccl_context_new_from_device_index", vp, ci, _E)
_sig("ccl_context_new_gpu", vp, _E)
_sig("ccl_context_new_offline", vp, _E)
_sig("clo_sort_get_key_spec", vp, vp)
_sig("clo_sort_get_jit", vp, vp)
_sig("ccl_context_destroy", None, vp)
_sig("ccl_context_get_device", vp, vp, _u32, _E)
_sig("ccl_device_get_index", ci, vp)
_sig("ccl_device_get_max_work_group_size", sz, vp)
_sig("ccl_device_get_name", C.c_char_p, vp)
_sig("ccl_queue_new", vp, vp, vp, C.c_uint64, _E)
_sig("ccl_queue_new_from_stream", vp, vp, vp, C.c_uint64, _E)
_sig("ccl_queue_destroy", None, vp)
_sig("ccl_queue_finish", _u32, vp, _E)
_sig("ccl_queue_gc", None, vp)
_sig("ccl_queue_get_stream", vp, vp)
_sig("ccl_buffer_new", vp, vp, C.c_uint64, sz, vp, _E)
_sig("ccl_buffer_new_from_device_ptr", vp, vp, vp, sz, _E)
_sig("ccl_buffer_destroy", None, vp)
_sig("ccl_buffer_get_size", sz, vp)
_sig("ccl_buffer_get_device_ptr", vp, vp)
_sig("ccl_buffer_enqueue_write", vp, vp, vp, _u32, sz, sz, vp, vp, _E)
_sig("ccl_buffer_enqueue_read", vp, vp, vp, _u32, sz, sz, vp, vp, _E)
_sig("ccl_buffer_enqueue_copy", vp, vp, vp, vp, sz, sz, sz, vp, _E)
_sig("ccl_event_get_name", C.c_char_p, vp)
_sig("ccl_event_wait", _u32, C.POINTER(vp), _E)
_sig("ccl_event_wait_list_clear", None, C.POINTER(vp))
lib.ccl_event_wait_list_add.restype = None      # variadic: (CCLEventWaitList*, CCLEvent*, ..., NULL)
_sig("ccl_prof_new", vp)
_sig("ccl_prof_destroy", None, vp)
_sig("ccl_prof_add_queue", None, vp, C.c_char_p, vp)
_sig("ccl_prof_calc", _u32, vp, _E)
_sig("ccl_prof_get_duration", C.c_uint64, vp)


class ProfAgg(C.Structure):
    _fields_ = [("event_name", C.c_char_p), ("absolute_time", C.c_uint64), ("relative_time", C.c_double)]


_sig("ccl_prof_get_agg", C.POINTER(ProfAgg), vp, C.c_char_p)
_sig("ccl_prof_iter_agg_init", None, vp, ci)
_sig("ccl_prof_iter_agg_next", C.POINTER(ProfAgg), vp)

_sig("clo_sort_new", vp, C.c_char_p, C.c_char_p, vp, C.POINTER(ci), C.POINTER(ci), C.c_char_p, C.c_char_p,
     C.c_char_p, _E)
_sig("clo_sort_destroy", None, vp)
_sig("clo_sort_with_device_data", vp, vp, vp, vp, vp, vp, sz, sz, _E)
_sig("clo_sort_with_host_data", _u32, vp, vp, vp, vp, vp, sz, sz, _E)
_sig("clo_sort_get_context", vp, vp)
_sig("clo_sort_get_program", vp, vp)
_sig("clo_sort_get_element_type", ci, vp)
_sig("clo_sort_get_element_size", sz, vp)
_sig("clo_sort_get_key_type", ci, vp)
_sig("clo_sort_get_key_size", sz, vp)
_sig("clo_sort_get_data", vp, vp)
_sig("clo_sort_set_data", None, vp, vp)
_sig("clo_sort_get_num_kernels", _u32, vp, _E)
_sig("clo_sort_get_kernel_name", C.c_char_p, vp, _u32, _E)
_sig("clo_sort_get_localmem_usage", sz, vp, _u32, sz, sz, _E)

_sig("clo_scan_new", vp, C.c_char_p, C.c_char_p, vp, ci, ci, C.c_char_p, _E)
_sig("clo_scan_destroy", None, vp)
_sig("clo_scan_with_device_data", vp, vp, vp, vp, vp, vp, sz, sz, _E)
_sig("clo_scan_with_host_data", _u32, vp, vp, vp, vp, vp, sz, sz, _E)
_sig("clo_scan_get_context", vp, vp)
_sig("clo_scan_get_program", vp, vp)
_sig("clo_scan_get_elem_type", ci, vp)
_sig("clo_scan_get_element_size", sz, vp)
_sig("clo_scan_get_sum_type", ci, vp)
_sig("clo_scan_get_sum_size", sz, vp)
_sig("clo_scan_get_data", vp, vp)
_sig("clo_scan_set_data", None, vp, vp)
_sig("clo_scan_get_num_kernels", _u32, vp, _E)
_sig("clo_scan_get_kernel_name", C.c_char_p, vp, _u32, _E)
_sig("clo_scan_get_localmem_usage", sz, vp, _u32, sz, sz, _E)


# ---- sharded sort (include/clo_shard.h) ----
SHARD_AG = C.CFUNCTYPE(ci, vp, vp, vp, sz, vp)
SHARD_A2A = C.CFUNCTYPE(ci, vp, vp, C.POINTER(sz), C.POINTER(sz), vp, C.POINTER(sz), C.POINTER(sz), vp)
SHARD_DESTROY = C.CFUNCTYPE(None, vp)
SHARD_ALLOC = C.CFUNCTYPE(vp, vp, sz)
SHARD_FREE = C.CFUNCTYPE(None, vp, vp)
SHARD_ASYNC = C.CFUNCTYPE(ci, vp)


class ShardTransportStruct(C.Structure):
    _fields_ = [("user", vp), ("rank", ci), ("world", ci), ("all_gather_u64", SHARD_AG), ("all_to_all_v", SHARD_A2A),
                ("destroy", SHARD_DESTROY), ("abort", SHARD_DESTROY), ("recv_alloc", SHARD_ALLOC), ("recv_free", SHARD_FREE),
                ("async_error", SHARD_ASYNC)]


_sig("clo_shard_rccl_unique_id", _u32, vp, _E)
_sig("clo_shard_transport_new_rccl", C.POINTER(ShardTransportStruct), vp, ci, ci, _E)
_sig("clo_shard_transport_destroy", None, C.POINTER(ShardTransportStruct))
_sig("clo_shard_sort_new", vp, vp, C.POINTER(ShardTransportStruct), ci, C.c_char_p, _E)
_sig("clo_shard_sort_destroy", None, vp)
_sig("clo_shard_sort_with_device_data", vp, vp, vp, vp, sz, C.POINTER(vp), C.POINTER(sz), _E)
_sig("clo_shard_sort_get_phase_ms", None, vp, C.POINTER(C.c_double * 4))
_sig("clo_shard_sort_finish", _u32, vp, vp, C.c_uint, _E)
_sig("clo_shard_plan", None, C.POINTER(C.c_uint64), ci, ci, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz))
_sig("clo_shard_plan_slice", sz, C.POINTER(C.c_uint64), sz, ci, ci, ci, ci, ci, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz),
     C.POINTER(sz), C.POINTER(sz), C.POINTER(sz))
_sig("clo_shard_sort_get_exchange", None, vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_double), C.POINTER(ci))


class CloError(RuntimeError):
    """A GError reported by the library: .domain (string), .code, .message."""

    def __init__(self, domain, code, message):
        self.domain, self.code, self.message = domain, code, message
        super().__init__("[%s:%d] %s" % (domain, code, message))


class _Err:
    """GError** out-parameter holder."""

    def __init__(self):
        self.p = GErrorP()

    @property
    def ref(self):
        return C.byref(self.p)

    def raise_if_set(self):
        if self.p:
            e = self.p.contents
            dom = lib.clo_quark_to_string(e.domain)
            exc = CloError(dom.decode() if dom else str(e.domain), e.code,
                           e.message.decode() if e.message else "")
            lib.clo_gerror_free(self.p)
            self.p = GErrorP()
            raise exc


def _b(s):
    return None if s is None else s.encode()


def clo_type(t):
    """CloType constant from a name ('uint'), a numpy dtype, or an int."""
    if isinstance(t, int):
        return t
    if isinstance(t, str):
        return CLO_TYPES[t]
    return CLO_TYPES[_NP_TO_CLO[np.dtype(t)]]


class Context:
    """CCLContext over one HIP device."""

    def __init__(self, device_index=0, offline=False):
        err = _Err()
        if offline:  # host-logic tests only: nothing can be enqueued on it
            self.h = lib.ccl_context_new_offline(err.ref)
        else:
            self.h = lib.ccl_context_new_from_device_index(device_index, err.ref)
        err.raise_if_set()
        if not self.h:
            raise CloError("clo", CLO_ERROR_LIBRARY, "could not create context")
        self.device_index = device_index

    @property
    def device(self):
        return lib.ccl_context_get_device(self.h, 0, None)

    @property
    def device_name(self):
        n = lib.ccl_device_get_name(self.device)
        return n.decode() if n else ""

    def close(self):
        if self.h:
            lib.ccl_context_destroy(self.h)
            self.h = None


def _make_ewl(events):
    """A CCLEventWaitList (an opaque pointer, NULL when empty) holding `events`."""
    ewl = vp(None)
    if events is None:
        return ewl
    if not isinstance(events, (list, tuple)):
        events = [events]
    for e in events:
        if e:
            lib.ccl_event_wait_list_add(C.byref(ewl), vp(e), vp(None))
    return ewl


def wait_for_events(events):
    """ccl_event_wait: host-side wait for every event of the list; raises CloError if a command
    behind one of them failed (a look-back give-up is reported here too)."""
    ewl = _make_ewl(events)
    if not ewl:
        return
    err = _Err()
    lib.ccl_event_wait(C.byref(ewl), err.ref)
    err.raise_if_set()


class Queue:
    """CCLQueue = one HIP stream (own, or adopted from e.g. torch)."""

    def __init__(self, ctx, profiling=False, stream=None):
        err = _Err()
        props = CL_QUEUE_PROFILING_ENABLE if profiling else 0
        if stream is None:
            self.h = lib.ccl_queue_new(ctx.h, None, props, err.ref)
        else:
            self.h = lib.ccl_queue_new_from_stream(ctx.h, vp(stream), props, err.ref)
        err.raise_if_set()
        self.ctx = ctx

    @property
    def stream(self):
        return lib.ccl_queue_get_stream(self.h)

    def finish(self):
        err = _Err()
        lib.ccl_queue_finish(self.h, err.ref)
        err.raise_if_set()

    def gc(self):
        lib.ccl_queue_gc(self.h)

    def close(self):
        if self.h:
            lib.ccl_queue_destroy(self.h)
            self.h = None


class Buffer:
    """CCLBuffer: device memory (owned, or wrapping an external device pointer)."""

    def __init__(self, ctx, nbytes=None, device_ptr=None):
        err = _Err()
        if device_ptr is None:
            self.h = lib.ccl_buffer_new(ctx.h, 1, nbytes, None, err.ref)
        else:
            self.h = lib.ccl_buffer_new_from_device_ptr(ctx.h, vp(device_ptr), nbytes, err.ref)
        err.raise_if_set()
        self.ctx = ctx
        self.nbytes = nbytes

    @property
    def ptr(self):
        return lib.ccl_buffer_get_device_ptr(self.h)

    def write(self, queue, array, offset=0):
        a = np.ascontiguousarray(array)
        err = _Err()
        lib.ccl_buffer_enqueue_write(self.h, queue.h, 1, offset, a.nbytes, a.ctypes.data_as(vp), None, err.ref)
        err.raise_if_set()

    def read(self, queue, dtype, count, offset=0, wait_for=None):
        """Blocking read; `wait_for`: an event (or a list of events) of any queue the copy has to come after."""
        out = np.empty(count, dtype=dtype)
        err = _Err()
        ewl = _make_ewl(wait_for)
        lib.ccl_buffer_enqueue_read(self.h, queue.h, 1, offset, out.nbytes, out.ctypes.data_as(vp),
                                    C.byref(ewl) if ewl else None, err.ref)
        if ewl:
            lib.ccl_event_wait_list_clear(C.byref(ewl))
        err.raise_if_set()
        return out

    def close(self):
        if self.h:
            lib.ccl_buffer_destroy(self.h)
            self.h = None


class Sorter:
    """CloSort. Arguments as clo_sort_new (clo_sort_abstract.in.h:116-120)."""

    def __init__(self, algorithm, ctx, elem_type, key_type=None, options=None, compare=None, get_key=None,
                 compiler_opts=None):
        et = ci(clo_type(elem_type))
        kt = ci(clo_type(key_type)) if key_type is not None else None
        err = _Err()
        self.h = lib.clo_sort_new(_b(algorithm), _b(options), ctx.h, C.byref(et),
                                  C.byref(kt) if kt is not None else None, _b(compare), _b(get_key),
                                  _b(compiler_opts), err.ref)
        err.raise_if_set()
        if not self.h:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_sort_new returned NULL")
        self.ctx = ctx

    def with_device_data(self, q_exec, data_in, data_out, numel, lws_max=0, q_comm=None):
        err = _Err()
        evt = lib.clo_sort_with_device_data(self.h, q_exec.h, q_comm.h if q_comm else None, data_in.h,
                                            data_out.h if data_out is not None else None, numel, lws_max, err.ref)
        err.raise_if_set()
        return evt

    def with_host_data(self, array, q_exec=None, q_comm=None, lws_max=0):
        a = np.ascontiguousarray(array)
        out = np.empty_like(a)
        err = _Err()
        ok = lib.clo_sort_with_host_data(self.h, q_exec.h if q_exec else None, q_comm.h if q_comm else None,
                                         a.ctypes.data_as(vp), out.ctypes.data_as(vp), a.size, lws_max, err.ref)
        err.raise_if_set()
        if not ok:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_sort_with_host_data failed")
        return out

    def key_spec(self):
        """(elem_size, key_size, key_shift, key_bits, key_kind, descending) as parsed from
        elem/key types, get_key and compare."""
        p = C.cast(lib.clo_sort_get_key_spec(self.h), C.POINTER(C.c_int * 6))
        return tuple(p.contents)

    @property
    def element_size(self):
        return lib.clo_sort_get_element_size(self.h)

    @property
    def key_size(self):
        return lib.clo_sort_get_key_size(self.h)

    @property
    def element_type(self):
        return lib.clo_sort_get_element_type(self.h)

    @property
    def key_type(self):
        return lib.clo_sort_get_key_type(self.h)

    def num_kernels(self):
        err = _Err()
        n = lib.clo_sort_get_num_kernels(self.h, err.ref)
        err.raise_if_set()
        return n

    def kernel_name(self, i):
        err = _Err()
        n = lib.clo_sort_get_kernel_name(self.h, i, err.ref)
        err.raise_if_set()
        return n.decode() if n else None

    def localmem_usage(self, i, lws_max=0, numel=1 << 20):
        err = _Err()
        n = lib.clo_sort_get_localmem_usage(self.h, i, lws_max, numel, err.ref)
        err.raise_if_set()
        return n

    def close(self):
        if self.h:
            lib.clo_sort_destroy(self.h)
            self.h = None


class Scanner:
    """CloScan. Arguments as clo_scan_new (clo_scan_abstract.in.h:109-112)."""

    def __init__(self, algorithm, ctx, elem_type, sum_type, options=None, compiler_opts=None):
        err = _Err()
        self.h = lib.clo_scan_new(_b(algorithm), _b(options), ctx.h, clo_type(elem_type), clo_type(sum_type),
                                  _b(compiler_opts), err.ref)
        err.raise_if_set()
        if not self.h:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_scan_new returned NULL")
        self.ctx = ctx
        self.sum_np = CLO_TYPE_NP[[k for k, v in CLO_TYPES.items() if v == clo_type(sum_type)][0]]

    def with_device_data(self, q_exec, data_in, data_out, numel, lws_max=0, q_comm=None):
        err = _Err()
        evt = lib.clo_scan_with_device_data(self.h, q_exec.h, q_comm.h if q_comm else None, data_in.h,
                                            data_out.h, numel, lws_max, err.ref)
        err.raise_if_set()
        return evt

    def with_host_data(self, array, q_exec=None, q_comm=None, lws_max=0):
        a = np.ascontiguousarray(array)
        out = np.empty(a.size, dtype=self.sum_np)
        err = _Err()
        ok = lib.clo_scan_with_host_data(self.h, q_exec.h if q_exec else None, q_comm.h if q_comm else None,
                                         a.ctypes.data_as(vp), out.ctypes.data_as(vp), a.size, lws_max, err.ref)
        err.raise_if_set()
        if not ok:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_scan_with_host_data failed")
        return out

    def num_kernels(self):
        return lib.clo_scan_get_num_kernels(self.h, None)

    def kernel_name(self, i):
        n = lib.clo_scan_get_kernel_name(self.h, i, None)
        return n.decode() if n else None

    def localmem_usage(self, i, lws_max=0, numel=1 << 20):
        return lib.clo_scan_get_localmem_usage(self.h, i, lws_max, numel, None)

    def close(self):
        if self.h:
            lib.clo_scan_destroy(self.h)
            self.h = None


class ShardTransport:
    """CloShardTransport: RCCL over xGMI (`ShardTransport.rccl(id, rank, world)`), or two Python
    callables moving the same bytes (`ShardTransport.custom(rank, world, all_gather, all_to_all_v)`:
    tests with two ranks on one GPU, where RCCL cannot be used)."""

    def __init__(self, ptr, keep=None, owned=True):
        self.ptr, self._keep, self._owned = ptr, keep, owned

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        err = _Err()
        lib.clo_shard_rccl_unique_id(buf, err.ref)
        err.raise_if_set()
        return bytes(buf)

    @classmethod
    def rccl(cls, unique_id, rank, world):
        err = _Err()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        p = lib.clo_shard_transport_new_rccl(buf, rank, world, err.ref)
        err.raise_if_set()
        if not p:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_shard_transport_new_rccl returned NULL")
        return cls(p)

    @classmethod
    def custom(cls, rank, world, all_gather_u64, all_to_all_v, recv_alloc=None, recv_free=None, async_error=None):
        """all_gather_u64(send_ptr, recv_ptr, count, stream) -> status;
        all_to_all_v(send_ptr, send_bytes, send_off, recv_ptr, recv_bytes, recv_off, stream) -> status
        (the four arrays as Python lists of `world` ints); optional recv_alloc(bytes) -> device pointer or None,
        recv_free(ptr): the transport's own memory for the receive buffers; async_error() -> 0 or a status, polled by the
        sort's bounded waits."""
        def ag(user, s, r, count, stream):
            return int(all_gather_u64(s, r, count, stream) or 0)

        def a2a(user, s, sb, so, r, rb, ro, stream):
            return int(all_to_all_v(s, [sb[i] for i in range(world)], [so[i] for i in range(world)], r,
                                    [rb[i] for i in range(world)], [ro[i] for i in range(world)], stream) or 0)
        aborted = []

        def ab(user):
            aborted.append(True)
        def ra(user, nbytes):
            return recv_alloc(nbytes) or None

        def rf(user, ptr):
            recv_free(ptr)

        def ae(user):
            return int(async_error() or 0)
        st = ShardTransportStruct(None, rank, world, SHARD_AG(ag), SHARD_A2A(a2a), SHARD_DESTROY(), SHARD_DESTROY(ab),
                                  SHARD_ALLOC(ra) if recv_alloc else SHARD_ALLOC(), SHARD_FREE(rf) if recv_free else SHARD_FREE(),
                                  SHARD_ASYNC(ae) if async_error else SHARD_ASYNC())
        t = cls(C.pointer(st), keep=(st, ag, a2a, ab, ra, rf, ae), owned=False)
        t.aborted = aborted          # non-empty once the C driver has asked for an abort
        return t

    def close(self):
        if self.ptr and self._owned:
            lib.clo_shard_transport_destroy(self.ptr)
        self.ptr = None


class ShardSort:
    """CloShardSort (include/clo_shard.h): MSD bucket exchange + local satradix behind the C API."""

    def __init__(self, ctx, transport, elem_type, options=None):
        err = _Err()
        self.h = lib.clo_shard_sort_new(ctx.h, transport.ptr, clo_type(elem_type), _b(options), err.ref)
        err.raise_if_set()
        if not self.h:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_shard_sort_new returned NULL")
        self.ctx, self.transport = ctx, transport

    def with_device_data(self, q_exec, data_in, numel):
        """-> (device pointer of this rank's sorted bucket, its length); the memory belongs to the object."""
        out, m = vp(), sz(0)
        err = _Err()
        lib.clo_shard_sort_with_device_data(self.h, q_exec.h, data_in.h if data_in is not None else None, numel,
                                            C.byref(out), C.byref(m), err.ref)
        err.raise_if_set()
        return lib.ccl_buffer_get_device_ptr(out), m.value

    def finish(self, q_exec, timeout_ms=0):
        """Bounded wait for what the last call left running (clo_shard_sort_finish); raises CloError when the time is up or
        the transport has failed — this rank's side of the transport is aborted then."""
        err = _Err()
        ok = lib.clo_shard_sort_finish(self.h, q_exec.h, int(timeout_ms), err.ref)
        err.raise_if_set()
        if not ok:
            raise CloError("clo", CLO_ERROR_LIBRARY, "clo_shard_sort_finish failed")

    def phase_ms(self):
        a = (C.c_double * 4)()
        lib.clo_shard_sort_get_phase_ms(self.h, C.byref(a))
        return dict(zip(("partition", "count_exchange", "key_exchange", "local_sort"), a))

    def exchange(self):
        """The key exchange of the last call: bytes sent to / received from other ranks, device ms from the first
        all-to-all(v) to the end of the last, number of slices used."""
        bo, bi, ms, sl = sz(0), sz(0), C.c_double(0), ci(0)
        lib.clo_shard_sort_get_exchange(self.h, C.byref(bo), C.byref(bi), C.byref(ms), C.byref(sl))
        return {"bytes_out": bo.value, "bytes_in": bi.value, "ms": ms.value, "slices": sl.value}

    def close(self):
        if self.h:
            lib.clo_shard_sort_destroy(self.h)
            self.h = None


class Profiler:
    """CCLProf over queues created with profiling=True (clo_sort_bench.c:201-208)."""

    def __init__(self, *queues):
        self.h = lib.ccl_prof_new()
        for q in queues:
            lib.ccl_prof_add_queue(self.h, b"q", q.h)

    def duration_ns(self):
        err = _Err()
        lib.ccl_prof_calc(self.h, err.ref)
        err.raise_if_set()
        return lib.ccl_prof_get_duration(self.h)

    def aggregates(self):
        """{event name: total ns} of the last duration_ns() call (cf4ocl2: ccl_prof_iter_agg_*)."""
        out = {}
        lib.ccl_prof_iter_agg_init(self.h, 0)
        while True:
            a = lib.ccl_prof_iter_agg_next(self.h)
            if not a:
                break
            out[a.contents.event_name.decode()] = a.contents.absolute_time
        return out

    def close(self):
        if self.h:
            lib.ccl_prof_destroy(self.h)
            self.h = None


class HipEventTimer:
    """Pair of HIP events on a queue's stream (the stream the kernels run on)."""

    def __init__(self, queue):
        self.q = queue
        self.e0, self.e1 = vp(), vp()
        _hip.check(lib.clo_hip_event_create(C.byref(self.e0)))
        _hip.check(lib.clo_hip_event_create(C.byref(self.e1)))

    def start(self):
        _hip.check(lib.clo_hip_event_record(self.e0, self.q.stream))

    def stop(self):
        _hip.check(lib.clo_hip_event_record(self.e1, self.q.stream))

    def elapsed_ms(self):
        _hip.check(lib.clo_hip_event_synchronize(self.e1))
        ms = C.c_float()
        _hip.check(lib.clo_hip_event_elapsed_ms(self.e0, self.e1, C.byref(ms)))
        return ms.value

    def close(self):
        lib.clo_hip_event_destroy(self.e0)
        lib.clo_hip_event_destroy(self.e1)
