"""ctypes bindings for the CPU oracle (oracle/libclo_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libclo_oracle.so")

KEY_UNSIGNED, KEY_SIGNED, KEY_FLOAT = 0, 1, 2


class Desc(C.Structure):
    _fields_ = [("elem_size", C.c_int), ("key_size", C.c_int), ("key_shift", C.c_int),
                ("key_kind", C.c_int), ("descending", C.c_int)]


def build():
    src = os.path.join(_ORACLE_DIR, "clo_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
        L.clo_oracle_nlpo2.restype = C.c_uint
        L.clo_oracle_nlpo2.argtypes = [C.c_uint]
        L.clo_oracle_ones32.restype = C.c_uint
        L.clo_oracle_ones32.argtypes = [C.c_uint]
        L.clo_oracle_tzc.restype = C.c_uint
        L.clo_oracle_tzc.argtypes = [C.c_int]
        L.clo_oracle_sbitonic.argtypes = [vp, sz, C.POINTER(Desc)]
        L.clo_oracle_sbitonic.restype = None
        L.clo_oracle_gselect.argtypes = [vp, vp, sz, C.POINTER(Desc)]
        L.clo_oracle_gselect.restype = None
        L.clo_oracle_abitonic.argtypes = [vp, sz, C.POINTER(Desc), sz, sz, C.c_uint, C.c_uint, C.c_uint]
        L.clo_oracle_abitonic.restype = C.c_int
        L.clo_oracle_satradix.argtypes = [vp, sz, C.POINTER(Desc), C.c_uint, sz, sz, vp, vp, vp]
        L.clo_oracle_satradix.restype = C.c_int
        L.clo_oracle_blelloch.argtypes = [vp, vp, sz, C.c_int, C.c_int, sz, sz]
        L.clo_oracle_blelloch.restype = C.c_int
        L.clo_oracle_serial_scan.argtypes = [vp, vp, sz, C.c_int, C.c_int]
        L.clo_oracle_serial_scan.restype = None
        L.clo_oracle_check_sorted.argtypes = [vp, sz, C.c_int, C.c_int]
        L.clo_oracle_check_sorted.restype = C.c_long
        L.clo_oracle_stable_sort.argtypes = [vp, sz, C.POINTER(Desc)]
        L.clo_oracle_stable_sort.restype = None
        L.clo_oracle_bench_rand.argtypes = [C.c_uint32, C.c_int, vp, sz]
        L.clo_oracle_bench_rand.restype = None
        L.clo_oracle_scan_bench_rand.argtypes = [C.c_uint32, C.c_int, vp, sz]
        L.clo_oracle_scan_bench_rand.restype = None
        L.clo_oracle_sbitonic_mt.argtypes = [vp, sz, C.POINTER(Desc), C.c_int]
        L.clo_oracle_sbitonic_mt.restype = None
        L.clo_oracle_abitonic_mt.argtypes = [vp, sz, C.POINTER(Desc), sz, sz, C.c_uint, C.c_uint, C.c_uint, C.c_int]
        L.clo_oracle_abitonic_mt.restype = C.c_int
        L.clo_oracle_satradix_mt.argtypes = [vp, sz, C.POINTER(Desc), C.c_uint, sz, C.c_int]
        L.clo_oracle_satradix_mt.restype = C.c_int
        L.clo_oracle_blelloch_mt.argtypes = [vp, vp, sz, C.c_int, C.c_int, sz, C.c_int]
        L.clo_oracle_blelloch_mt.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def desc_for(arr, key_size=None, key_shift=0, key_kind=KEY_UNSIGNED, descending=False):
    es = arr.dtype.itemsize
    return Desc(es, key_size or es, key_shift, key_kind, int(descending))


def sbitonic(arr, threads=1, **kw):
    """threads > 1 (0: all host cores): the pairs of every step spread over OpenMP threads, same bits."""
    out = np.ascontiguousarray(arr).copy()
    d = desc_for(out, **kw)
    if threads == 1:
        lib().clo_oracle_sbitonic(_p(out), out.size, C.byref(d))
    else:
        lib().clo_oracle_sbitonic_mt(_p(out), out.size, C.byref(d), threads)
    return out


def gselect(arr, **kw):
    a = np.ascontiguousarray(arr)
    out = np.empty_like(a)
    d = desc_for(a, **kw)
    lib().clo_oracle_gselect(_p(a), _p(out), a.size, C.byref(d))
    return out


def abitonic(arr, lws_max=0, dev_max_lws=256, minps=1, maxps=4, maxsfs=0xFFFFFFFF, threads=1, **kw):
    out = np.ascontiguousarray(arr).copy()
    d = desc_for(out, **kw)
    if threads == 1:
        launches = lib().clo_oracle_abitonic(_p(out), out.size, C.byref(d), lws_max, dev_max_lws,
                                             minps, maxps, maxsfs)
    else:
        launches = lib().clo_oracle_abitonic_mt(_p(out), out.size, C.byref(d), lws_max, dev_max_lws,
                                                minps, maxps, maxsfs, threads)
    return out, launches


def satradix(arr, radix=16, lws_max=0, dev_max_lws=256, debug=False, threads=1, **kw):
    out = np.ascontiguousarray(arr).copy()
    d = desc_for(out, **kw)
    if threads != 1:
        r = lib().clo_oracle_satradix_mt(_p(out), out.size, C.byref(d), radix,
                                         lws_max or dev_max_lws, threads)
        if r < 0:
            raise ValueError("oracle satradix_mt failed: %d" % r)
        return out
    if debug:
        n = lib().clo_oracle_nlpo2(out.size)
        L = max(min(lws_max or dev_max_lws, dev_max_lws, n), radix)
        naux = (n // L) * radix
        offs = np.zeros(naux, np.uint32)
        cnt = np.zeros(naux, np.uint32)
        cs = np.zeros(naux, np.uint32)
        r = lib().clo_oracle_satradix(_p(out), out.size, C.byref(d), radix, lws_max, dev_max_lws,
                                      _p(offs), _p(cnt), _p(cs))
        if r < 0:
            raise ValueError("oracle satradix failed: %d" % r)
        return out, offs, cnt, cs
    r = lib().clo_oracle_satradix(_p(out), out.size, C.byref(d), radix, lws_max, dev_max_lws,
                                  None, None, None)
    if r < 0:
        raise ValueError("oracle satradix failed: %d" % r)
    return out


def blelloch(arr, sum_dtype, lws_max=0, dev_max_lws=256, threads=1):
    a = np.ascontiguousarray(arr)
    out = np.zeros(a.size, dtype=sum_dtype)
    if threads != 1:
        lib().clo_oracle_blelloch_mt(_p(a), _p(out), a.size, a.dtype.itemsize, out.dtype.itemsize,
                                     lws_max or dev_max_lws, threads)
    else:
        lib().clo_oracle_blelloch(_p(a), _p(out), a.size, a.dtype.itemsize, out.dtype.itemsize,
                                  lws_max, dev_max_lws)
    return out


def serial_scan(arr, sum_dtype):
    a = np.ascontiguousarray(arr)
    out = np.zeros(a.size, dtype=sum_dtype)
    lib().clo_oracle_serial_scan(_p(a), _p(out), a.size, a.dtype.itemsize, out.dtype.itemsize)
    return out


def check_sorted(arr, kind=KEY_UNSIGNED):
    a = np.ascontiguousarray(arr)
    return lib().clo_oracle_check_sorted(_p(a), a.size, a.dtype.itemsize, kind)


def stable_sort(arr, **kw):
    out = np.ascontiguousarray(arr).copy()
    d = desc_for(out, **kw)
    lib().clo_oracle_stable_sort(_p(out), out.size, C.byref(d))
    return out


_CLO_TYPES = {"char": (0, np.int8), "uchar": (1, np.uint8), "short": (2, np.int16),
              "ushort": (3, np.uint16), "int": (4, np.int32), "uint": (5, np.uint32),
              "long": (6, np.int64), "ulong": (7, np.uint64)}


def bench_rand(seed, type_name, numel):
    t, dt = _CLO_TYPES[type_name]
    out = np.zeros(numel, dtype=dt)
    lib().clo_oracle_bench_rand(seed, t, _p(out), numel)
    return out


def scan_bench_rand(seed, dtype, numel):
    out = np.zeros(numel, dtype=dtype)
    lib().clo_oracle_scan_bench_rand(seed, out.dtype.itemsize, _p(out), numel)
    return out
