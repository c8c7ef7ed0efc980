"""Distinct sorter / scanner objects used from distinct host threads at the same time, each on its own queue, one shared
context: the threading contract of the reference (no locks, no mutable globals: an object is not re-entrant, distinct
objects are independent — SURVEY §8b "Threading"). ctypes drops the GIL during a call, so the C drivers, their workspace
caches and the run-time compiler really do run side by side here. Every result is checked bit for bit against numpy."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clo():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cl_ops_amd
    return cl_ops_amd


def _sort_worker(clo, ctx, alg, etype, seed, rounds, max_log2, errors, options=None, **kw):
    try:
        rng = np.random.default_rng(seed)
        dt = clo.api.CLO_TYPE_NP[etype]
        q = clo.Queue(ctx)
        s = clo.Sorter(alg, ctx, etype, options=options, **kw)
        info = np.iinfo(dt)
        for _ in range(rounds):
            n = int(rng.integers(1, 1 << int(rng.integers(4, max_log2 + 1))))
            a = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64 if info.min < 0 else np.uint64).astype(dt)
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
            src.write(q, a)
            s.with_device_data(q, src, dst, n)
            got = dst.read(q, dt, n)
            if not np.array_equal(got, np.sort(a)):
                errors.append("%s %s n=%d: wrong result" % (alg, etype, n))
            src.close()
            dst.close()
        s.close()
        q.close()
    except Exception as e:  # noqa: BLE001 (a worker thread must hand its failure to the test)
        errors.append("%s %s: %r" % (alg, etype, e))


def _scan_worker(clo, ctx, etype, stype, seed, rounds, max_log2, errors):
    try:
        rng = np.random.default_rng(seed)
        dt, sdt = clo.api.CLO_TYPE_NP[etype], clo.api.CLO_TYPE_NP[stype]
        q = clo.Queue(ctx)
        sc = clo.Scanner("blelloch", ctx, etype, stype)
        for _ in range(rounds):
            n = int(rng.integers(1, 1 << int(rng.integers(4, max_log2 + 1))))
            a = rng.integers(0, 128, n).astype(dt)
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, n * sdt.itemsize)
            src.write(q, a)
            sc.with_device_data(q, src, dst, n)
            got = dst.read(q, sdt, n)
            exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))[:-1])).astype(sdt)
            if not np.array_equal(got, exp):
                errors.append("scan %s->%s n=%d: wrong result" % (etype, stype, n))
            src.close()
            dst.close()
        sc.close()
        q.close()
    except Exception as e:  # noqa: BLE001
        errors.append("scan %s->%s: %r" % (etype, stype, e))


def test_distinct_objects_on_distinct_threads(clo):
    ctx = clo.Context(0)
    errors = []
    workers = [
        threading.Thread(target=_sort_worker, args=(clo, ctx, "satradix", "uint", 11, 24, 23, errors)),
        threading.Thread(target=_sort_worker, args=(clo, ctx, "satradix", "ulong", 12, 20, 22, errors)),
        threading.Thread(target=_sort_worker, args=(clo, ctx, "satradix", "ushort", 13, 24, 21, errors), kwargs={"options": "radix=256"}),
        threading.Thread(target=_sort_worker, args=(clo, ctx, "abitonic", "int", 14, 16, 21, errors)),
        threading.Thread(target=_sort_worker, args=(clo, ctx, "sbitonic", "uint", 15, 16, 15, errors)),
        threading.Thread(target=_scan_worker, args=(clo, ctx, "uint", "uint", 16, 30, 23, errors)),
        threading.Thread(target=_scan_worker, args=(clo, ctx, "uint", "ulong", 17, 30, 22, errors)),
    ]
    for w in workers:
        w.start()
    for w in workers:
        w.join(timeout=600)
    assert not any(w.is_alive() for w in workers), "a worker is stuck"
    assert errors == []
    ctx.close()


def test_sorters_built_by_the_runtime_compiler_side_by_side(clo):
    """Two threads construct sorters through hiprtc at once (different key expressions) and sort with them."""
    ctx = clo.Context(0)
    errors = []

    def worker(shift, seed):
        try:
            rng = np.random.default_rng(seed)
            q = clo.Queue(ctx)
            for alg in ("satradix", "abitonic"):
                s = clo.Sorter(alg, ctx, "uint", get_key="((x) >> SHIFT) & 0xfff", compiler_opts="-DSHIFT=%d" % shift)
                n = 1 << 16
                a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
                src.write(q, a)
                s.with_device_data(q, src, dst, n)
                got = dst.read(q, np.uint32, n)
                k = (got >> np.uint32(shift)) & np.uint32(0xfff)
                if not (np.all(k[:-1] <= k[1:]) and np.array_equal(np.sort(got), np.sort(a))):
                    errors.append("%s shift=%d: wrong result" % (alg, shift))
                src.close()
                dst.close()
                s.close()
            q.close()
        except Exception as e:  # noqa: BLE001
            errors.append("shift=%d: %r" % (shift, e))

    workers = [threading.Thread(target=worker, args=(sh, 20 + sh)) for sh in (4, 12, 20)]
    for w in workers:
        w.start()
    for w in workers:
        w.join(timeout=600)
    assert not any(w.is_alive() for w in workers), "a worker is stuck"
    assert errors == []
    ctx.close()


def test_host_data_pipelines_side_by_side(clo):
    """The host-buffer entry points above their pipelining thresholds (helper threads, chunked copies, the sort's
    bucket-by-bucket segmented sorts), three objects at once, each on a queue of its own."""
    ctx = clo.Context(0)
    errors = []

    def sort_worker(etype, log2n, seed):
        try:
            dt = clo.api.CLO_TYPE_NP[etype]
            rng = np.random.default_rng(seed)
            a = rng.integers(0, np.iinfo(dt).max, 1 << log2n, dtype=np.uint64).astype(dt)
            q = clo.Queue(ctx)
            s = clo.Sorter("satradix", ctx, etype)
            for _ in range(2):
                got = s.with_host_data(a, q_exec=q)
                if not np.array_equal(got, np.sort(a)):
                    errors.append("sort %s 2^%d: wrong result" % (etype, log2n))
            s.close()
            q.close()
        except Exception as e:  # noqa: BLE001
            errors.append("sort %s: %r" % (etype, e))

    def scan_worker(log2n, seed):
        try:
            a = np.random.default_rng(seed).integers(0, 128, 1 << log2n).astype(np.uint32)
            q = clo.Queue(ctx)
            sc = clo.Scanner("blelloch", ctx, "uint", "ulong")
            exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))[:-1]))
            for _ in range(2):
                got = sc.with_host_data(a, q_exec=q)
                if not np.array_equal(got, exp):
                    errors.append("scan 2^%d: wrong result" % log2n)
            sc.close()
            q.close()
        except Exception as e:  # noqa: BLE001
            errors.append("scan: %r" % (e,))

    workers = [threading.Thread(target=sort_worker, args=("uint", 25, 31)), threading.Thread(target=sort_worker, args=("ulong", 24, 32)),
               threading.Thread(target=scan_worker, args=(26, 33))]
    for w in workers:
        w.start()
    for w in workers:
        w.join(timeout=600)
    assert not any(w.is_alive() for w in workers), "a worker is stuck"
    assert errors == []
    ctx.close()


def test_objects_give_their_device_memory_back(clo):
    """Sorters, scanners and their queues created, used and destroyed over and over: the device's free memory comes back
    to where it was (cached workspaces, pipeline buffers, events and streams all belong to an object and go with it)."""
    import torch
    ctx = clo.Context(0)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 128, 1 << 21).astype(np.uint32)

    def cycle():
        q = clo.Queue(ctx)
        for alg, opts in (("satradix", None), ("satradix", "radix=256"), ("abitonic", None), ("sbitonic", None)):
            s = clo.Sorter(alg, ctx, "uint", options=opts)
            n = a.size if alg != "sbitonic" else 1 << 15
            src, dst = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4)
            src.write(q, a[:n])
            s.with_device_data(q, src, dst, n)
            assert np.array_equal(dst.read(q, np.uint32, n), np.sort(a[:n]))
            if alg == "satradix":
                assert np.array_equal(s.with_host_data(a, q), np.sort(a))
            for x in (src, dst, s):
                x.close()
        sc = clo.Scanner("blelloch", ctx, "uint", "ulong")
        got = sc.with_host_data(b, q)
        assert int(got[-1]) == int(b[:-1].sum())
        sc.close()
        q.close()

    cycle()                       # (first use: code objects, the runtime's own pools)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (64 << 20), "device memory lost over 12 cycles: %.1f MiB" % ((free0 - free1) / 2**20)
    ctx.close()
