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
    from cl_ops_amd._hip import lib
    monkeypatch.setenv("CLO_MAX_SPINS", "0")
    lib.clo_hip_env_refresh()                 # (the switches are read when an object is made — or on request)
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
    lib.clo_hip_env_refresh()
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
    n = 1 << 24
    a = np.random.default_rng(0).integers(0, 1 << 32, n, dtype=np.uint32)
    src = clo.Buffer(ctx, n * 4)
    s = clo.Sorter("satradix", ctx, "uint")
    # 2^24 elements: chain-free passes (histogram, counter scan, scatter per digit pair);
    # 2^20: single-sweep passes (one up-front histogram, then scatters that hand the
    # counts from tile to tile: no counter-scan kernel); 2^12: one launch sorts it all
    for m, names in ((n, {"satradix_histogram", "clo_scan_blelloch_wgscan", "satradix_scatter"}),
                     (1 << 20, {"satradix_histogram", "satradix_scatter"}), (1 << 12, {"satradix_localsort"})):
        src.write(qp, a[:m])
        clo.Profiler(qp).duration_ns()          # (drop the copy's event)
        s.with_device_data(qp, src, None, m)
        prof = clo.Profiler(qp)
        total = prof.duration_ns()
        agg = prof.aggregates()
        assert set(agg) == names, (m, agg)
        assert sum(agg.values()) == total and all(v > 0 for v in agg.values())
        assert np.array_equal(src.read(qp, np.uint32, m), np.sort(a[:m]))
        prof.close()
    s.close()
    prof = clo.Profiler(qp)

    n = 1 << 20
    sc = clo.Scanner("blelloch", ctx, "uint", "uint")
    dst = clo.Buffer(ctx, n * 4)
    sc.with_device_data(qp, src, dst, n)
    prof2 = clo.Profiler(qp)
    prof2.duration_ns()                      # (includes the read above)
    assert "clo_scan_blelloch_wgscan" in prof2.aggregates()
    sc.close()

    for steps in ("1", None):                     # one launch per step, then the default tiled schedule: one event name either way
        if steps:
            os.environ["CLO_SBITONIC_STEPS"] = steps
        else:
            os.environ.pop("CLO_SBITONIC_STEPS", None)
        try:
            s = clo.Sorter("sbitonic", ctx, "uint")
            m = 1 << 16
            src.write(qp, a[:m])
            clo.Profiler(qp).duration_ns()
            s.with_device_data(qp, src, None, m)
            s.with_device_data(qp, src, None, m)   # a repeat: no graph replay on a profiling queue
            prof3 = clo.Profiler(qp)
            prof3.duration_ns()
            assert set(prof3.aggregates()) == {"sbitonic_ndrange"}
            assert np.array_equal(src.read(qp, np.uint32, m), np.sort(a[:m]))
            s.close()
            if steps:
                prof3.close()
        finally:
            os.environ.pop("CLO_SBITONIC_STEPS", None)
    for x in (prof, prof2, prof3, src, dst, qp):
        x.close()


@pytest.mark.parametrize("steps", [True, False])
def test_sbitonic_repeats_on_an_uncapturable_stream(gpu, monkeypatch, steps):
    """Graph replay (the one-launch-per-step schedule, CLO_SBITONIC_STEPS=1) is best effort: the
    legacy NULL stream (a queue adopted from torch's default stream) cannot be captured —
    repeated identical sorts on it must still sort. The default tiled schedule beside it."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    if steps:
        monkeypatch.setenv("CLO_SBITONIC_STEPS", "1")
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


@pytest.mark.parametrize("dist", ["all equal", "8 distinct", "90 % in one bin of every digit", "sorted", "quarter of each wave equal"])
def test_sweep_passes_on_repeating_keys(gpu, dist):
    """Sizes the library sorts with the single-sweep passes, keys that repeat: waves whose
    lanes agree count their digits by groups (one ballot per group) instead of one LDS add
    per lane; the result must not depend on which way a wave counted. Keys alone and
    stable (key, index) pairs."""
    import cl_ops_amd as clo
    ctx, q = gpu
    n = (1 << 20) + 77
    rng = np.random.default_rng(len(dist))
    if dist == "all equal":
        k = np.full(n, 0x5A5A5A5A, np.uint32)
    elif dist == "8 distinct":
        k = (rng.integers(0, 8, n, dtype=np.uint32) * np.uint32(0x11111111)).astype(np.uint32)
    elif dist.startswith("90"):
        k = np.where(rng.integers(0, 10, n) > 0, np.uint32(0x77777777), rng.integers(0, 1 << 32, n, dtype=np.uint32)).astype(np.uint32)
    elif dist == "sorted":
        k = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint32))
    else:   # lanes 0..15 of every wave's first element agree, everything else is random
        k = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        first = np.arange(0, n, 16)            # element 0 of every thread (16 consecutive elements per thread)
        lanes = (first // 16) % 64
        k[first[lanes < 16]] = 0xC3C3C3C3
    s = clo.Sorter("satradix", ctx, "uint")
    assert np.array_equal(s.with_host_data(k, q), np.sort(k))
    s.close()
    pairs = (k.astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    sp = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")
    assert np.array_equal(sp.with_host_data(pairs, q), O.stable_sort(pairs, key_size=4, key_shift=32))
    sp.close()


@pytest.mark.parametrize("case", ["uint radix 16", "uint radix 256", "uint radix 4", "uint radix 64", "ulong radix 16", "pairs radix 16",
                                  "uint radix 16 no-digit-stream", "ulong radix 16 no-digit-stream"])
def test_big_tiles_ragged_sizes(gpu, monkeypatch, case):
    """Arrays of 256 MiB and more (64 MiB of 8-byte elements) run the chain-free passes
    on 16 384-element tiles (1024 threads, the table of ends inside the stage), every
    pass but the last also writing the digit stream the next histogram reads: sizes
    that end inside a tile, every digit-width family, 4- and 8-byte elements, stable
    pairs."""
    import cl_ops_amd as clo
    ctx, q = gpu
    if case.endswith("no-digit-stream"):          # histograms over the elements in every pass
        monkeypatch.setenv("CLO_RADIX_NO_DIGITS", "1")
        case = case.rsplit(" ", 1)[0]
    et, _, radix = case.split()
    rng = np.random.default_rng(len(case))
    if et == "uint":
        n = (1 << 26) + 16384 + 4099
        a = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        s = clo.Sorter("satradix", ctx, "uint", options="radix=" + radix)
        exp = np.sort(a)
    elif et == "ulong":
        n = (1 << 25) + 8192 + 77
        a = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
        s = clo.Sorter("satradix", ctx, "ulong", options="radix=" + radix)
        exp = np.sort(a)
    else:
        n = (1 << 25) + 8192 + 77
        a = (rng.integers(0, 1 << 20, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)   # ~32 duplicates per key
        s = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)", options="radix=" + radix)
        exp = O.stable_sort(a, key_size=4, key_shift=32)
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("case", [("uint", "uint", "(x) * 2654435761u", lambda a: a * np.uint32(2654435761)),
                                  ("ulong", "ulong", "(x) * 0x9E3779B97F4A7C15ul", lambda a: a * np.uint64(0x9E3779B97F4A7C15))])
def test_jit_get_key_on_big_tiles(gpu, case):
    """A run-time compiled get_key at a size where the (key, index) pairs it sorts fill 256
    MiB and more: 16 384-element tiles, digit stream; 8-byte keys take two rounds."""
    import cl_ops_amd as clo
    ctx, q = gpu
    et, kt, expr, fn = case
    dt = clo.api.CLO_TYPE_NP[et]
    n = (1 << 25) + 12345
    a = np.random.default_rng(25).integers(0, np.iinfo(dt).max, n, dtype=np.uint64).astype(dt)
    s = clo.Sorter("satradix", ctx, et, key_type=kt, get_key=expr)
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, a[np.argsort(fn(a), kind="stable")])


def test_sorter_and_scanner_move_between_queues(gpu):
    """The cached buffers of a sorter / scanner follow the queue of the call: two
    live queues alternating (the later call waits for the earlier one's work, no
    host synchronisation in between), and a queue destroyed while the sorter
    still remembers it (its struct is held, its stream is gone)."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    n = (1 << 22) + 3
    rng = np.random.default_rng(11)
    q1, q2 = clo.Queue(ctx), clo.Queue(ctx)
    s = clo.Sorter("satradix", ctx, "uint")
    sc = clo.Scanner("blelloch", ctx, "uint", "ulong")
    bufs = []
    for k in range(6):
        q = (q1, q2)[k % 2]
        a = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        src, dst, sums = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 8)
        src.write(q, a)
        s.with_device_data(q, src, dst, n)     # no finish: the next call, on the other queue, reuses the sorter's buffers
        sc.with_device_data(q, src, sums, n)
        bufs.append((q, a, src, dst, sums))
    q1.finish()
    q2.finish()
    for q, a, src, dst, sums in bufs:
        assert np.array_equal(dst.read(q, np.uint32, n), np.sort(a))
        assert np.array_equal(sums.read(q, np.uint64, n), _excl(a, np.uint64))
        for x in (src, dst, sums):
            x.close()
    q2.close()                                  # the sorter's last call ran on q2
    q3 = clo.Queue(ctx)
    a = rng.integers(0, 1 << 32, n, dtype=np.uint32)
    assert np.array_equal(s.with_host_data(a, q3), np.sort(a))
    assert np.array_equal(sc.with_host_data(a, q3), _excl(a, np.uint64))
    for x in (s, sc, q1, q3):
        x.close()


@pytest.mark.parametrize("path", ["library's choice", "chain-free pair passes"])
def test_full_size_satradix_u64_2p28_shard_of_config_5(gpu, monkeypatch, path):
    """Config 5's per-GPU shard: 2^28 uint64 keys, radix 16. Size-independent
    properties on the full array (order, multiset checksums, idempotence of a
    second sort) and exact equality with numpy on slices picked by value. The
    library sorts this size with the single-sweep passes; the other path too."""
    import torch
    import cl_ops_amd as clo
    ctx, q = gpu
    if path != "library's choice":
        monkeypatch.setenv("CLO_RADIX_SWEEP", "0")
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


# ----------------------------------------------------------------------------
# two ranks on the one GPU of this box, config 5's element type at scale, skewed
# ----------------------------------------------------------------------------

def _skewed_u64_worker(rank, world, port, log2n, out_dir):
    import torch
    import torch.distributed as dist
    from cl_ops_amd.multigpu import HipLocalOps
    from numpy_ops import host_staged_sorter_class
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n = 1 << log2n
        rng = np.random.default_rng(500 + rank)
        a = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
        # 85 % of BOTH ranks' keys get the top bit set: rank 1 receives ~1.7 n keys,
        # more than capacity_factor (1.25) provides for -> the count-exact reallocation
        force = rng.random(n) < 0.7
        a[force] |= np.uint64(1) << np.uint64(63)
        local = torch.from_numpy(a.view(np.int64)).cuda()
        ops = HipLocalOps("ulong", 0)
        ss = host_staged_sorter_class()(ops)
        out, m = ss.sort(local, n)
        torch.cuda.synchronize()
        got = out[:m].cpu().numpy().view(np.uint64)
        ok_sorted = bool(np.all(got[:-1] <= got[1:]))
        ok_bucket = bool(np.all((got >> np.uint64(63)) == rank))
        stats = np.array([m, int(got.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(got)),
                          int(a.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(a)),
                          int(ok_sorted), int(ok_bucket), int(m > int(n * ss.capacity_factor) + 1024)], dtype=np.uint64)
        np.save(os.path.join(out_dir, "stats_%d.npy" % rank), stats)
        # exact check of one slice by value: keys whose top 16 bits are (rank << 15) | 0x1234
        top = np.uint64((rank << 15) | 0x1234)
        np.save(os.path.join(out_dir, "slice_in_%d.npy" % rank), a[(a >> np.uint64(48)) == top])
        np.save(os.path.join(out_dir, "slice_out_%d.npy" % rank), got[(got >> np.uint64(48)) == top])
        other = np.uint64(((1 - rank) << 15) | 0x1234)
        np.save(os.path.join(out_dir, "slice_for_peer_%d.npy" % rank), a[(a >> np.uint64(48)) == other])
        assert np.array_equal(local.cpu().numpy().view(np.uint64), a)   # the shard is only read
        ops.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_skewed_u64_2p27_each_on_one_gpu(gpu, tmp_path):
    """MSD partition -> exchange plan -> local satradix with 2^27 uint64 keys per rank and a
    bucket that overflows the receive capacity; the all-to-all alone is staged through the
    host (RCCL refuses two ranks on one device)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    log2n = 27
    mp.spawn(_skewed_u64_worker, args=(2, port, log2n, str(tmp_path)), nprocs=2, join=True)
    st = [np.load(tmp_path / ("stats_%d.npy" % r)) for r in range(2)]
    n = 1 << log2n
    assert int(st[0][0]) + int(st[1][0]) == 2 * n                                  # nothing lost
    assert (int(st[0][1]) + int(st[1][1])) % (1 << 64) == (int(st[0][3]) + int(st[1][3])) % (1 << 64)
    assert int(st[0][2]) ^ int(st[1][2]) == int(st[0][4]) ^ int(st[1][4])
    assert all(int(s_[5]) == 1 and int(s_[6]) == 1 for s_ in st)                  # sorted, and rank r holds bucket r
    assert int(st[1][7]) == 1, "the skew did not exceed the receive capacity: %d keys" % int(st[1][0])
    for r in range(2):
        exp = np.sort(np.concatenate([np.load(tmp_path / ("slice_in_%d.npy" % r)),
                                      np.load(tmp_path / ("slice_for_peer_%d.npy" % (1 - r)))]))
        assert np.array_equal(np.load(tmp_path / ("slice_out_%d.npy" % r)), exp)


# ----------------------------------------------------------------------------
# the sharded sort behind the C API (include/clo_shard.h)
# ----------------------------------------------------------------------------

def test_c_shard_sort_world_one_over_rccl(gpu):
    """One rank: RCCL communicator of size 1 (id, init, destroy are the real calls), the
    sort is a copy + local satradix; the input is left untouched."""
    import torch
    import cl_ops_amd as clo
    from cl_ops_amd.multigpu import CShardedSorter
    for et, dt, tdt in (("uint", np.uint32, np.int32), ("ulong", np.uint64, np.int64)):
        a = np.random.default_rng(3).integers(0, np.iinfo(dt).max, 300007, dtype=dt, endpoint=True)
        t = torch.from_numpy(a.view(tdt).copy()).cuda()
        s = CShardedSorter(et, 0)
        out, m = s.sort(t)
        torch.cuda.synchronize()
        assert m == a.size and np.array_equal(out.cpu().numpy().view(dt), np.sort(a))
        assert np.array_equal(t.cpu().numpy().view(dt), a)
        s.close()


def _c_shard_worker(rank, world, port, n, out_dir):
    import ctypes as C
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    from cl_ops_amd.multigpu import CShardedSorter
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # a transport that moves the same bytes through host memory over gloo (RCCL
        # refuses two ranks on one device); everything else is the C code under test
        from shard_transport import gloo_staged_transport
        tr = gloo_staged_transport(rank, world)
        a = np.random.default_rng(70 + rank).integers(0, np.iinfo(np.uint64).max, n + 1000 * rank, dtype=np.uint64, endpoint=True)
        if rank == 1:
            a[: a.size // 2] |= np.uint64(1) << np.uint64(63)      # uneven buckets
        local = torch.from_numpy(a.view(np.int64).copy()).cuda()
        s = CShardedSorter("ulong", 0, transport=tr)
        for _ in range(2):                                          # the second call reuses every buffer
            out, m = s.sort(local)
            torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), a)
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), out.cpu().numpy().view(np.uint64)[:m])
        assert np.array_equal(local.cpu().numpy().view(np.uint64), a)
        s.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_c_shard_sort_two_ranks_on_one_gpu(gpu, tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_c_shard_worker, args=(2, port, 200000, str(tmp_path)), nprocs=2, join=True)
    ins = [np.load(tmp_path / ("in_%d.npy" % r)) for r in range(2)]
    outs = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(2)]
    assert np.array_equal(np.concatenate(outs), np.sort(np.concatenate(ins)))
    assert np.all(outs[0] >> np.uint64(63) == 0) and np.all(outs[1] >> np.uint64(63) == 1)


# ----------------------------------------------------------------------------
# floating-point scans (deterministic reduce / scan / apply: clo_hip_fscan.hip)
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("types", [("float", "float"), ("double", "double"), ("float", "double"), ("uint", "float"),
                                   ("int", "double"), ("half", "float"), ("double", "float")])
@pytest.mark.parametrize("n", [1, 17, 4095, 4096, 4097, (1 << 20) + 3, (1 << 24) + 4099 * 4096 + 5])
def test_float_scan_matches_a_float64_reference(gpu, types, n):
    """Sums in float / double: equal to the exact prefix sums to rounding (the order of the
    additions is the kernels' own tree, not upstream's), bit-identical from run to run, and
    exact where every partial sum is representable."""
    import cl_ops_amd as clo
    ctx, q = gpu
    et, st = types
    edt, sdt = clo.api.CLO_TYPE_NP[et], clo.api.CLO_TYPE_NP[st]
    rng = np.random.default_rng(n % 1000 + len(et))
    if np.issubdtype(edt, np.floating):
        a = (rng.random(n) - 0.25).astype(edt)
    else:
        a = rng.integers(-50 if np.issubdtype(edt, np.signedinteger) else 0, 128, n).astype(edt)
    sc = clo.Scanner("blelloch", ctx, et, st)
    got = sc.with_host_data(a, q)
    again = sc.with_host_data(a, q)
    sc.close()
    assert got.dtype == sdt and np.array_equal(got, again), "not deterministic"
    wide = a.astype(np.longdouble)     # (80-bit: the reference's own rounding stays far below a double's)
    exact = np.concatenate(([0.0], np.cumsum(wide)[:-1]))
    scale = np.concatenate(([0.0], np.cumsum(np.abs(wide))[:-1])) + 1.0
    eps = np.finfo(sdt).eps
    err = np.abs(got.astype(np.longdouble) - exact) / scale
    assert np.all(err <= 256 * eps), float(err.max() / eps)
    if not np.issubdtype(edt, np.floating) and n <= (1 << 17):
        assert np.array_equal(got.astype(np.int64), exact.astype(np.int64))     # small integers: every sum exact
