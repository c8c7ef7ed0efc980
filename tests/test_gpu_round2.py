"""GPU tests added in round 2: the scan's self-maintained workspace (epochs, the
wrap, the give-up report), per-kernel events on profiling queues, best-effort
graph replay, event recycling, and full-size checks of config 5's shard.
Everything goes through the C-ABI of libcl_ops_hip.so; expected values come
from numpy / the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _excl(a, sdt):
    wide = a.astype(np.uint64)
    return np.concatenate((np.zeros(1, np.uint64), np.cumsum(wide[:-1], dtype=np.uint64))).astype(sdt)


@pytest.mark.parametrize("types", [("uint", "uint"), ("uint", "ulong")])
def test_scan_workspace_is_never_cleared_between_calls(gpu, types):
    """One scanner, sizes going up and down (both kernel shapes, partial last
    super-tiles): a call clears nothing, so whatever an earlier call left in the
    workspace must read as 'not written' to the next one."""
    import cl_ops_amd as clo
    ctx, q = gpu
    et, st = types
    sdt = clo.api.CLO_TYPE_NP[st]
    sc = clo.Scanner("blelloch", ctx, et, st)
    rng = np.random.default_rng(5)
    sizes = [(1 << 24) + 77, 1000, (1 << 22) + 5, 1 << 25, 300001, (1 << 24) + 77, 16384 * 64 + 1, 1 << 20, 5, 1 << 24]
    nmax = max(sizes)
    src, dst = clo.Buffer(ctx, nmax * 4), clo.Buffer(ctx, nmax * sdt.itemsize)
    for k, n in enumerate(sizes):
        a = rng.integers(0, 1 << 32 if k % 3 == 0 else 128, n, dtype=np.uint32)
        src.write(q, a)
        sc.with_device_data(q, src, dst, n)
        q.finish()
        assert np.array_equal(dst.read(q, sdt, n), _excl(a, sdt)), "call %d, n=%d" % (k, n)
    sc.close()
    src.close()
    dst.close()


@pytest.mark.parametrize("sum_size", [4, 8])
def test_scan_epoch_wrap(gpu, sum_size):
    """Epochs run 1 .. 2^30-1, then the last work-group to leave zeroes the
    workspace and they start over. Forced here with the test hook."""
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    ctx, q = gpu
    sdt = np.dtype(np.uint32 if sum_size == 4 else np.uint64)
    nmax = (1 << 24) + 4097
    wsb = lib.clo_hip_scan_workspace_bytes(nmax, 4, sum_size)
    ws, src, dst = clo.Buffer(ctx, wsb), clo.Buffer(ctx, nmax * 4), clo.Buffer(ctx, nmax * sum_size)
    _hip.check(lib.clo_hip_scan_workspace_init(ws.ptr, wsb, q.stream))
    rng = np.random.default_rng(sum_size)
    # a big scan leaves entries all over the workspace; then jump to just before the wrap
    sizes = [nmax, 70000, nmax, 1 << 22, 70000, nmax, 1 << 20, nmax]
    for k, n in enumerate(sizes):
        if k == 1:
            _hip.check(lib.clo_hip_scan_workspace_set_epoch(ws.ptr, (1 << 30) - 4, q.stream))
        a = rng.integers(0, 1 << 16, n, dtype=np.uint32)
        src.write(q, a)
        _hip.check(lib.clo_hip_scan_exclusive(src.ptr, dst.ptr, n, 4, 0, sum_size, ws.ptr, wsb, q.stream))
        q.finish()
        assert lib.clo_hip_check_status(ws.ptr, q.stream) == 0
        assert np.array_equal(dst.read(q, sdt, n), _excl(a, sdt)), "call %d, n=%d" % (k, n)
    hdr = ws.read(q, np.uint32, 4)
    assert 1 <= int(hdr[2]) <= 8, "the epoch word did not wrap: %d" % hdr[2]
    for b in (ws, src, dst):
        b.close()


def test_scan_lookback_timeout_is_reported(gpu, monkeypatch):
    """A look-back that gives up (forced: CLO_MAX_SPINS=0 lets a tile wait for
    nobody) must not pass as success: clo_scan_with_host_data fails with
    CLO_ERROR_LIBRARY, ccl_queue_finish after clo_scan_with_device_data too, and
    the scanner works again afterwards."""
    import cl_ops_amd as clo
    ctx, q = gpu
    n = 1 << 25
    a = np.random.default_rng(9).integers(0, 128, n, dtype=np.uint32)
    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    assert np.array_equal(sc.with_host_data(a, q), _excl(a, np.uint32))
    monkeypatch.setenv("CLO_MAX_SPINS", "0")
    with pytest.raises(clo.CloError) as e:
        sc.with_host_data(a, q)
    assert e.value.code == clo.api.CLO_ERROR_LIBRARY and "look-back" in e.value.message
    src, dst = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4)
    src.write(q, a)
    sc.with_device_data(q, src, dst, n)
    with pytest.raises(clo.CloError) as e:
        q.finish()
    assert e.value.code == clo.api.CLO_ERROR_LIBRARY
    monkeypatch.delenv("CLO_MAX_SPINS")
    sc.with_device_data(q, src, dst, n)
    q.finish()
    assert np.array_equal(dst.read(q, np.uint32, n), _excl(a, np.uint32))
    assert np.array_equal(sc.with_host_data(a, q), _excl(a, np.uint32))
    for x in (sc, src, dst):
        x.close()


def test_profiling_queue_gets_one_event_per_kernel(gpu):
    """Upstream names the event of every launch (clo_sort_satradix.c:282,295,312;
    clo_scan_blelloch.c:158; clo_sort_sbitonic.c:115) and CCLProf aggregates by
    name; a queue created with CL_QUEUE_PROFILING_ENABLE sees the same names here."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    qp = clo.Queue(ctx, profiling=True)
    n = 1 << 20
    a = O.bench_rand(0, "uint", n)
    src = clo.Buffer(ctx, n * 4)
    src.write(qp, a)
    clo.Profiler(qp).duration_ns()          # (drop the copy's event)

    s = clo.Sorter("satradix", ctx, "uint")
    s.with_device_data(qp, src, None, n)
    prof = clo.Profiler(qp)
    total = prof.duration_ns()
    agg = prof.aggregates()
    assert set(agg) == {"satradix_histogram", "clo_scan_blelloch_wgscan", "satradix_scatter"}, agg
    assert sum(agg.values()) == total and all(v > 0 for v in agg.values())
    assert np.array_equal(src.read(qp, np.uint32, n), np.sort(a))
    s.close()

    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    dst = clo.Buffer(ctx, n * 4)
    sc.with_device_data(qp, src, dst, n)
    prof2 = clo.Profiler(qp)
    prof2.duration_ns()                      # (includes the read above)
    assert "clo_scan_blelloch_wgscan" in prof2.aggregates()
    sc.close()

    s = clo.Sorter("sbitonic", ctx, "uint")
    src.write(qp, a[:1 << 10])
    clo.Profiler(qp).duration_ns()
    s.with_device_data(qp, src, None, 1 << 10)
    s.with_device_data(qp, src, None, 1 << 10)   # a repeat: no graph replay on a profiling queue
    prof3 = clo.Profiler(qp)
    prof3.duration_ns()
    assert set(prof3.aggregates()) == {"sbitonic_ndrange"}
    assert np.array_equal(src.read(qp, np.uint32, 1 << 10), np.sort(a[:1 << 10]))
    s.close()
    for x in (prof, prof2, prof3, src, dst, qp):
        x.close()


def test_sbitonic_repeats_on_an_uncapturable_stream(gpu):
    """Graph replay is best effort: the legacy NULL stream (a queue adopted from
    torch's default stream) cannot be captured — repeated identical sorts on it
    must still sort."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    q0 = clo.Queue(ctx, stream=0)
    n = 1 << 12
    buf = clo.Buffer(ctx, n * 4)
    s = clo.Sorter("sbitonic", ctx, "uint")
    for k in range(4):
        a = O.bench_rand(k, "uint", n)
        buf.write(q0, a)
        s.with_device_data(q0, buf, None, n)
        q0.finish()
        assert np.array_equal(buf.read(q0, np.uint32, n), np.sort(a)), "repeat %d" % k
    for x in (s, buf, q0):
        x.close()


def test_queue_without_profiling_recycles_its_events(gpu):
    """A loop of sorts on one queue must not pile up events without bound (and
    must keep working while old events are dropped)."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    q = clo.Queue(ctx)
    n = 1 << 12
    a = O.bench_rand(1, "uint", n)
    src, dst = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4)
    src.write(q, a)
    s = clo.Sorter("satradix", ctx, "uint")
    for _ in range(1000):
        s.with_device_data(q, src, dst, n)
    q.finish()
    assert np.array_equal(dst.read(q, np.uint32, n), np.sort(a))
    for x in (s, src, dst, q):
        x.close()


def test_full_size_satradix_u64_2p28_shard_of_config_5(gpu):
    """Config 5's per-GPU shard: 2^28 uint64 keys, radix 16. Size-independent
    properties on the full array (order, multiset checksums, idempotence of a
    second sort) and exact equality with numpy on slices picked by value."""
    import torch
    import cl_ops_amd as clo
    ctx, q = gpu
    n = 1 << 28
    g = torch.Generator(device="cuda")
    g.manual_seed(28)
    hi = torch.randint(0, 1 << 32, (n,), generator=g, device="cuda", dtype=torch.int64)
    lo = torch.randint(0, 1 << 32, (n,), generator=g, device="cuda", dtype=torch.int64)
    src = (hi << 32) | lo            # full 64 random bits, as int64 bit patterns
    del hi, lo
    dst = torch.empty_like(src)
    bsrc = clo.Buffer(ctx, n * 8, device_ptr=src.data_ptr())
    bdst = clo.Buffer(ctx, n * 8, device_ptr=dst.data_ptr())
    torch.cuda.synchronize()
    s = clo.Sorter("satradix", ctx, "ulong")
    s.with_device_data(q, bsrc, bdst, n)
    q.finish()
    # unsigned order of int64 bit patterns: flip the sign bit
    flip = torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
    d = dst ^ flip
    assert bool((d[1:] >= d[:-1]).all()), "not sorted"
    del d
    assert int(src.sum()) == int(dst.sum()), "sum of keys changed"            # (mod 2^64)
    x_in = src[0].clone()
    x_out = dst[0].clone()
    for t, acc in ((src, x_in), (dst, x_out)):
        # xor-fold in halves (torch has no xor reduction)
        v = t
        while v.numel() > 1:
            h = v.numel() // 2
            v = v[:h] ^ v[h:2 * h]
        acc.copy_(v[0])
    assert int(x_in) == int(x_out), "xor of keys changed"
    # exact: all keys with the top 12 bits == p, against numpy, for three p
    u = dst.cpu().numpy().view(np.uint64)
    for p in (0, 0x7FF, 0xFFF):
        lo_v, hi_v = np.uint64(p) << np.uint64(52), (np.uint64(p + 1) << np.uint64(52)) if p < 0xFFF else None
        sel = ((src >> 52) & 0xFFF) == p
        exp = np.sort(src[sel].cpu().numpy().view(np.uint64))
        a0 = np.searchsorted(u, lo_v, side="left")
        a1 = np.searchsorted(u, hi_v, side="left") if hi_v is not None else n
        assert a1 - a0 == exp.size and np.array_equal(u[a0:a1], exp), "bucket %#x differs" % p
    del u
    # in place, on sorted input: nothing moves
    before = dst.clone()
    s.with_device_data(q, bdst, None, n)
    q.finish()
    assert torch.equal(before, dst)
    for x in (s, bsrc, bdst):
        x.close()
