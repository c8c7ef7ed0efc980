"""GPU tests added in round 3: a look-back give-up inside a SORT must never pass as
success (whatever queues the caller synchronises with), ticket pools of the
single-sweep passes, floating-point scans of mixed magnitudes, the scan workspace's
self-description, recycled events. Everything goes through the C-ABI of
libcl_ops_hip.so; expected values come from numpy / the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------
# give-ups of the single-sweep radix passes (the library's path for 2^15 .. 2^22 keys)
# ----------------------------------------------------------------------------

def test_sort_lookback_timeout_is_reported_on_every_queue_shape(gpu, monkeypatch):
    """CLO_MAX_SPINS=0 makes every tile whose predecessors have not published give up.
    Reference call shapes (benchmarks/clo_sort_bench.c:160-162,196): host data with one
    queue, host data with a separate transfer queue (the final wait then synchronises
    with THAT queue only), device data + ccl_queue_finish, device data + a read on
    another queue that waits for the sort's event. The sorter works again afterwards."""
    import cl_ops_amd as clo
    from cl_ops_amd._hip import lib
    ctx, q = gpu
    n = 1 << 20
    assert lib.clo_hip_radix_polls(n, 4, 4) == 1, "2^20 uint32 keys are expected on the single-sweep passes"
    a = O.bench_rand(3, "uint", n)
    want = np.sort(a)
    s = clo.Sorter("satradix", ctx, "uint")
    qx, qc = clo.Queue(ctx, profiling=True), clo.Queue(ctx)
    assert np.array_equal(s.with_host_data(a, qx, qc), want)

    monkeypatch.setenv("CLO_MAX_SPINS", "0")
    lib.clo_hip_env_refresh()                 # (the switches are read when an object is made — or on request)
    for queues in ((q, None), (qx, qc)):
        with pytest.raises(clo.CloError) as e:
            s.with_host_data(a, *queues)
        assert e.value.code == clo.api.CLO_ERROR_LIBRARY and "spin" in e.value.message, e.value.message
    src, dst = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4)
    src.write(q, a)
    s.with_device_data(qx, src, dst, n)
    with pytest.raises(clo.CloError) as e:
        qx.finish()
    assert e.value.code == clo.api.CLO_ERROR_LIBRARY
    # a read on ANOTHER queue that waits for the sort: the watch travels with the wait
    evt = s.with_device_data(qx, src, dst, n)
    with pytest.raises(clo.CloError) as e:
        dst.read(qc, np.uint32, n, wait_for=evt)
        qc.finish()
    assert e.value.code == clo.api.CLO_ERROR_LIBRARY
    monkeypatch.setenv("CLO_MAX_SPINS", str(1 << 16))
    lib.clo_hip_env_refresh()
    try:
        qx.finish()      # (the give-up above may still be pending on the exec queue: reported once more at most)
    except clo.CloError:
        pass
    s.with_device_data(qx, src, dst, n)
    qx.finish()
    assert np.array_equal(dst.read(qc, np.uint32, n), want)
    assert np.array_equal(s.with_host_data(a, qx, qc), want)
    for x in (s, src, dst, qx, qc):
        x.close()


def test_sort_harness_reports_a_give_up(monkeypatch):
    """The C harness (upstream's queue setup: a profiling exec queue + a transfer queue)
    must not print a rate for a sort whose kernels gave up."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "benchmarks"), "-s"])
    env = dict(os.environ, CLO_MAX_SPINS="0")
    r = subprocess.run([os.path.join(ROOT, "benchmarks", "bin", "clo_hip_sort_bench"), "-a", "satradix", "-t", "uint",
                        "-m", "20", "-n", "20", "-r", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and ("spin" in r.stderr or "did not work" in r.stdout), r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "benchmarks", "bin", "clo_hip_sort_bench"), "-a", "satradix", "-t", "uint",
                        "-m", "20", "-n", "20", "-r", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "did not work" not in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("pools", ["1", "8"])
@pytest.mark.parametrize("kind,logn", [("uint", 15), ("uint", 20), ("uint", 23), ("ulong", 21)])
def test_sweep_ticket_pools(gpu, monkeypatch, pools, kind, logn):
    """One ticket counter (the default: forward progress whatever the device looks
    like) and eight per-XCD pools give the same, correct order."""
    import cl_ops_amd as clo
    ctx, q = gpu
    monkeypatch.setenv("CLO_RADIX_SWEEP", "1")            # (the library itself leaves this path above 256 tiles)
    monkeypatch.setenv("CLO_R1_POOLS", pools)
    n = (1 << logn) + 12345
    a = O.bench_rand(logn, kind, n)
    s = clo.Sorter("satradix", ctx, kind)
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, np.sort(a))


@pytest.mark.parametrize("kind,n,opts", [("uint", (1 << 22) + 12345, None), ("uint", (1 << 26) + 999, None), ("ulong", (1 << 23) + 5, None),
                                         ("uint", 8192 * 2 + 1, None), ("uint", 8192 * 129 + 3, None), ("uint", (1 << 23) + 1, "radix=4"),
                                         ("ulong", (1 << 22) + 7, "radix=256"), ("uint", 8192 * 300 + 11, "radix=2"),
                                         ("uint", 8192 * 700 + 5, "radix=8"), ("uint", 8192 * 300 + 1, "radix=32"),
                                         ("ulong", 4096 * 520 + 3, "radix=64"), ("uint", 8192 * 260 + 9, "radix=128")])
def test_counter_scan_in_one_launch(gpu, monkeypatch, kind, n, opts):
    """The chain-free passes' counter scan (ticketed chunks, published chunk sums, digit bases in
    a row of their own): 2, 3, 129+ tiles, many chunks, both tile shapes, radix 4 / 16 / 256."""
    import cl_ops_amd as clo
    ctx, q = gpu
    monkeypatch.setenv("CLO_RADIX_SWEEP", "0")
    logn = int(np.ceil(np.log2(n)))
    a = O.bench_rand(logn, kind, n)
    s = clo.Sorter("satradix", ctx, kind, options=opts)
    got = s.with_host_data(a, q)
    again = s.with_host_data(a[::-1].copy(), q)          # (the same workspace: hand-off words zeroed by the histogram)
    s.close()
    assert np.array_equal(got, np.sort(a))
    assert np.array_equal(again, np.sort(a))


def test_two_sorters_on_two_queues_at_once(gpu):
    """Two sorts in flight on two streams: the counter scans' tickets and hand-off words live in
    each sorter's own workspace, and no work-group waits for one that has not started."""
    import cl_ops_amd as clo
    ctx, q = gpu
    q2 = clo.Queue(ctx)
    n1, n2 = (1 << 25) + 3, (1 << 24) + 77
    a1, a2 = O.bench_rand(25, "uint", n1), O.bench_rand(24, "ulong", n2)
    s1, s2 = clo.Sorter("satradix", ctx, "uint"), clo.Sorter("satradix", ctx, "ulong")
    b1, o1 = clo.Buffer(ctx, a1.nbytes), clo.Buffer(ctx, a1.nbytes)
    b2, o2 = clo.Buffer(ctx, a2.nbytes), clo.Buffer(ctx, a2.nbytes)
    b1.write(q, a1)
    b2.write(q2, a2)
    q.finish()
    q2.finish()
    for _ in range(6):                       # (nothing synchronises in between: the kernels of both interleave)
        s1.with_device_data(q, b1, o1, n1)
        s2.with_device_data(q2, b2, o2, n2)
    q.finish()
    q2.finish()
    assert np.array_equal(o1.read(q, np.uint32, n1), np.sort(a1))
    assert np.array_equal(o2.read(q2, np.uint64, n2), np.sort(a2))
    for x in (s1, s2, b1, o1, b2, o2, q2):
        x.close()


# ----------------------------------------------------------------------------
# floating-point scans: the exclusive offset of a thread is never `inclusive - own`
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("st", ["float", "double"])
def test_float_scan_of_mixed_magnitudes(gpu, st):
    """A thread's 16-element sum far larger than the prefix in front of it must not
    swallow that prefix: [1]*16 followed by [big]*16 gives out[16] == 16 exactly
    (upstream's down-sweep never subtracts: clo_scan_blelloch.cl:112-140)."""
    import cl_ops_amd as clo
    ctx, q = gpu
    sdt = clo.api.CLO_TYPE_NP[st]
    big = 6.25e7 if st == "float" else 1e17          # 16 * big: the sum of one thread
    n = 3 * 4096 + 100
    a = np.ones(n, dtype=sdt)
    for start in range(16, n - 16, 32):
        a[start:start + 16] = big
    a[4096 + 48:4096 + 64] = -big                    # mixed signs too
    sc = clo.Scanner("blelloch", ctx, st, st)
    got = sc.with_host_data(a, q)
    sc.close()
    assert got[16] == 16.0, got[:40]
    wide = a.astype(np.longdouble)
    exact = np.concatenate(([0.0], np.cumsum(wide)[:-1]))
    scale = np.concatenate(([0.0], np.cumsum(np.abs(wide))[:-1])) + 1.0
    err = np.abs(got.astype(np.longdouble) - exact) / scale
    assert np.all(err <= 64 * np.finfo(sdt).eps), float(err.max() / np.finfo(sdt).eps)


# ----------------------------------------------------------------------------
# the scan workspace describes itself: size and initialisation are checked, not assumed
# ----------------------------------------------------------------------------

def test_scan_workspace_header_is_validated(gpu):
    """clo_hip_scan_exclusive on a workspace that was never initialised, or with a
    byte count other than the one it was initialised with (one workspace sized for the
    largest n, `clo_hip_scan_workspace_bytes(n)` passed per call), answers
    CLO_HIP_EWORKSPACE instead of returning wrong sums."""
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    ctx, q = gpu
    nmax, nsmall = (1 << 22) + 3, 1 << 18
    big, small = lib.clo_hip_scan_workspace_bytes(nmax, 4, 4), lib.clo_hip_scan_workspace_bytes(nsmall, 4, 4)
    assert big > small
    ws, src, dst = clo.Buffer(ctx, big), clo.Buffer(ctx, nmax * 4), clo.Buffer(ctx, nmax * 4)
    a = np.random.default_rng(1).integers(0, 128, nmax, dtype=np.uint32)
    src.write(q, a)
    junk = np.full(big // 4, 0xA5A5A5A5, dtype=np.uint32)
    ws.write(q, junk)
    q.finish()
    st = lib.clo_hip_scan_exclusive(src.ptr, dst.ptr, nmax, 4, 0, 4, ws.ptr, big, q.stream)
    assert st == _hip.CLO_HIP_EWORKSPACE, st
    _hip.check(lib.clo_hip_scan_workspace_init(ws.ptr, big, q.stream))
    _hip.check(lib.clo_hip_scan_exclusive(src.ptr, dst.ptr, nmax, 4, 0, 4, ws.ptr, big, q.stream))
    q.finish()
    exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a[:-1], dtype=np.uint64))).astype(np.uint32)
    assert np.array_equal(dst.read(q, np.uint32, nmax), exp)
    # the natural round-1 usage: same allocation, a smaller byte count per call
    st = lib.clo_hip_scan_exclusive(src.ptr, dst.ptr, nsmall, 4, 0, 4, ws.ptr, small, q.stream)
    assert st == _hip.CLO_HIP_EWORKSPACE, st
    # the full size with a smaller n is fine (sizes going up and down on one workspace)
    for m in (nsmall, nmax, 1000, nmax):
        _hip.check(lib.clo_hip_scan_exclusive(src.ptr, dst.ptr, m, 4, 0, 4, ws.ptr, big, q.stream))
        q.finish()
        assert np.array_equal(dst.read(q, np.uint32, m), exp[:m])
    for x in (ws, src, dst):
        x.close()


# ----------------------------------------------------------------------------
# events of a queue without profiling are recycled, never freed under the caller
# ----------------------------------------------------------------------------

def test_event_handles_stay_valid_across_many_commands(gpu):
    """A CCLEvent* handed out by clo_sort_with_device_data stays usable after hundreds of
    further commands on a queue without profiling (cf4ocl2: until the queue is collected):
    waiting for the whole list works and does not touch freed memory."""
    import cl_ops_amd as clo
    ctx, _ = gpu
    q = clo.Queue(ctx)
    n = 1 << 12
    a = O.bench_rand(5, "uint", n)
    src, dst = clo.Buffer(ctx, n * 4), clo.Buffer(ctx, n * 4)
    src.write(q, a)
    s = clo.Sorter("satradix", ctx, "uint")
    events = [s.with_device_data(q, src, dst, n) for _ in range(400)]
    clo.wait_for_events(events)
    assert np.array_equal(dst.read(q, np.uint32, n), np.sort(a))
    for x in (s, src, dst, q):
        x.close()


# ----------------------------------------------------------------------------
# the sharded sort behind the C API: slices, ranks that fail together
# ----------------------------------------------------------------------------

def _shard_worker(rank, world, port, etype, n, options, fail, out_dir):
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    import cl_ops_amd as clo
    from cl_ops_amd.multigpu import CShardedSorter
    from shard_transport import gloo_staged_transport
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        dt, tdt = (np.uint32, np.int32) if etype == "uint" else (np.uint64, np.int64)
        bits = 8 * np.dtype(dt).itemsize
        tr = gloo_staged_transport(rank, world, own_memory=bool(fail))
        a = np.random.default_rng(90 + rank).integers(0, np.iinfo(dt).max, n + 4099 * rank, dtype=dt, endpoint=True)
        if fail and fail[1] == 2:      # everything into the failing rank's bucket: its receive buffer has to grow
            a = (a >> dt(1)) | dt(fail[0] << (bits - 1))
        elif rank == 1:
            a[: a.size // 3] |= dt(1) << dt(bits - 1)              # uneven buckets
            a[a.size // 3: a.size // 2] &= dt((1 << (bits - 3)) - 1)   # and one sub-bucket much fuller than the others
        local = torch.from_numpy(a.view(tdt).copy()).cuda()
        s = CShardedSorter(etype, 0, transport=tr, options=(options + "," if options else "") + "slice_min=16777216")   # (slices from 16 MiB per rank on: the default, 256 MiB, is a size for a node)
        text = ""
        if fail:
            # a failure of ONE rank's own: stage 1 — its arguments are wrong (numel beyond its buffer), found before the
            # count exchange; stage 2 — its receive buffer has to grow after the plan and the memory is not there
            try:
                if fail[1] == 1 and rank == fail[0]:
                    small = clo.Buffer(s.ctx, 64, device_ptr=local.data_ptr())
                    try:
                        s.ss.with_device_data(s.queue, small, local.numel())
                    finally:
                        small.close()
                else:
                    if fail[1] == 2 and rank == fail[0]:
                        s.sort(local[:1000])               # (first call: the usual capacity, for 1000 keys)
                        torch.cuda.synchronize()
                        tr.alloc_state["fail"] = True
                    elif fail[1] == 2:
                        s.sort(local[:1000])
                        torch.cuda.synchronize()
                    s.sort(local)
                text = "NO ERROR"
            except clo.CloError as e:
                text = e.message
            tr.alloc_state["fail"] = False
            torch.cuda.synchronize()
        for _ in range(2):                                          # the second call reuses every buffer
            out, m = s.sort(local)
            torch.cuda.synchronize()
        x = s.ss.exchange()
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), a)
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), out.cpu().numpy().view(dt)[:m])
        with open(os.path.join(out_dir, "info_%d.txt" % rank), "w") as f:
            f.write("%d %d %d %d\n%s" % (x["slices"], x["bytes_out"], x["bytes_in"], int(bool(tr.aborted)), text))
        assert np.array_equal(local.cpu().numpy().view(dt), a)
        s.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_shard(tmp_path, etype, n, options, fail=None, world=2):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shard_worker, args=(world, port, etype, n, options, fail, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / ("in_%d.npy" % r)) for r in range(world)]
    outs = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)]
    infos = [(tmp_path / ("info_%d.txt" % r)).read_text().split("\n", 1) for r in range(world)]
    bits = 8 * ins[0].dtype.itemsize
    b = world.bit_length() - 1
    assert np.array_equal(np.concatenate(outs), np.sort(np.concatenate(ins)))
    for r in range(world):                      # rank r holds exactly bucket r
        assert np.all(outs[r] >> ins[0].dtype.type(bits - b) == r)
    return ins, outs, infos


@pytest.mark.parametrize("etype,n,options,slices", [("ulong", (1 << 22) + 77, None, 4), ("uint", (1 << 22) + 5, "slices=8", 8),
                                                    ("uint", (1 << 22) + 5, "slices=2,radix=256", 2), ("ulong", 1 << 22, "slices=1", 1),
                                                    ("uint", 50000, None, 1)])
def test_c_shard_sort_in_slices(gpu, tmp_path, etype, n, options, slices):
    """Two ranks on the one GPU, the exchange staged through gloo: sub-bucket j of every rank
    travels as its own all-to-all(v) and is sorted where it lands while the next one travels
    (include/clo_shard.h). Small arrays fall back to one exchange whatever `slices` says."""
    ins, outs, infos = _run_shard(tmp_path, etype, n, options)
    es = ins[0].dtype.itemsize
    for r in range(2):
        used, bytes_out, bytes_in, aborted = (int(v) for v in infos[r][0].split())
        assert used == slices and aborted == 0
        other = 1 - r
        top = ins[r] >> ins[r].dtype.type(8 * es - 1)
        assert bytes_out == int(np.count_nonzero(top == other)) * es           # exactly the keys of the other rank's bucket
        assert bytes_in == int(np.count_nonzero((ins[other] >> ins[other].dtype.type(8 * es - 1)) == r)) * es


@pytest.mark.parametrize("etype,options,slices", [("uint", None, 4), ("ulong", "slices=8", 8), ("uint", "slices=2", 2)])
def test_c_shard_sort_slices_over_real_rccl_world_one(gpu, monkeypatch, etype, options, slices):
    """One rank, but everything else as on a node: the RCCL communicator, the slices' grouped
    send/recv on the transfer stream, the events between it and the exec stream, the slice
    sorts in place. (RCCL refuses two ranks on one device: more ranks go through the staged
    transport above.)"""
    import torch
    from cl_ops_amd.multigpu import CShardedSorter
    dt, tdt = (np.uint32, np.int32) if etype == "uint" else (np.uint64, np.int64)
    n = (1 << 23) + 4099
    a = np.random.default_rng(17).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    t = torch.from_numpy(a.view(tdt).copy()).cuda()
    s = CShardedSorter(etype, 0, options=(options + "," if options else "") + "loopback=1,slice_min=16777216")   # (one rank alone would skip the exchange)
    for rep in range(2):                                   # (the second call reuses every buffer and event)
        out, m = s.sort(t)
        s.check()
        torch.cuda.synchronize()
        assert m == n and np.array_equal(out.cpu().numpy().view(dt), np.sort(a))
        x = s.ss.exchange()
        assert x["slices"] == slices and x["bytes_out"] == 0 and x["bytes_in"] == 0      # nothing leaves the only rank
    assert np.array_equal(t.cpu().numpy().view(dt), a)
    s.close()


@pytest.mark.parametrize("etype", ["uint", "ulong"])
@pytest.mark.parametrize("where", ["low", "high", "middle"])
def test_c_shard_sort_slices_without_keys(gpu, etype, where):
    """Keys that leave whole slices EMPTY — every key in the lowest, the highest or one middle quarter of the key
    space, so that one of the four slices carries everything: an empty slice has no sort of its own, and which of the
    two buffers holds the result must not depend on it (an empty LAST slice used to name the wrong buffer)."""
    import torch
    from cl_ops_amd.multigpu import CShardedSorter
    dt, tdt = (np.uint32, np.int32) if etype == "uint" else (np.uint64, np.int64)
    bits = 8 * np.dtype(dt).itemsize
    n = (1 << 23) + 77
    a = np.random.default_rng(23).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True) >> dt(2)
    a |= dt({"low": 0, "middle": 2, "high": 3}[where]) << dt(bits - 2)
    t = torch.from_numpy(a.view(tdt).copy()).cuda()
    s = CShardedSorter(etype, 0, options="slices=4,loopback=1,slice_min=16777216")
    for rep in range(2):
        out, m = s.sort(t)
        s.check()
        torch.cuda.synchronize()
        assert m == n and np.array_equal(out.cpu().numpy().view(dt), np.sort(a))
        assert s.ss.exchange()["slices"] == 4
    s.close()


@pytest.mark.parametrize("options,slices", [(None, 4), ("slices=8", 8)])
def test_c_shard_sort_four_ranks_on_one_gpu(gpu, tmp_path, options, slices):
    """Four ranks (2 bucket bits + 2 or 3 slice bits: the partition's two-split form, 4 and 5 bits) on the
    one GPU, 2^22 keys each, exchange staged through gloo."""
    ins, outs, infos = _run_shard(tmp_path, "uint", (1 << 22) + 11, options, world=4)
    for r in range(4):
        used, bytes_out, bytes_in, aborted = (int(v) for v in infos[r][0].split())
        assert used == slices and aborted == 0
        assert bytes_out == int(np.count_nonzero((ins[r] >> np.uint32(30)) != r)) * 4


@pytest.mark.parametrize("fail", [(1, 1), (0, 2)])
def test_c_shard_ranks_fail_together(gpu, tmp_path, fail):
    """A rank that fails on its own — before the count exchange (stage 1) or while growing its
    receive buffer after the plan (stage 2) — makes EVERY rank return an error, nobody enters the
    key exchange, nothing is aborted, and the same objects sort correctly right afterwards."""
    ins, outs, infos = _run_shard(tmp_path, "ulong", 300000, None, fail=fail)
    for r in range(2):
        aborted = int(infos[r][0].split()[3])
        text = infos[r][1]
        assert aborted == 0 and text != "NO ERROR", text
        if r != fail[0]:
            assert ("rank %d" % fail[0]) in text and "no rank sorted" in text, text


# ----------------------------------------------------------------------------
# clo_sort_with_host_data with the transfers overlapped
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("kind,n", [("uint", (1 << 24) + 12345), ("ulong", 1 << 24), ("pairs", (1 << 24) + 7), ("uint", 1 << 25),
                                    ("ulong", (1 << 25) + 13),    # (the last chunk is of the other tile shape than the full ones)
                                    ("narrow", (1 << 24) + 5)])   # (every key in ONE of the 256 sub-buckets)
@pytest.mark.parametrize("two_queues", [False, True])
def test_sort_host_data_pipelined_equals_blocking(gpu, monkeypatch, kind, n, two_queues):
    """With CLO_SORT_HOST_PIPELINE=1 satradix's host-data path splits every chunk by the top 8 key bits
    as it arrives and sorts / copies out bucket by bucket (16 buckets of 16 sub-buckets, one segmented sort each); the result is upstream's blocking path's
    (sort/clo_sort_abstract.c:348-395) bit for bit, also for key/value pairs (stable)."""
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(n % 97)
    if kind == "uint":
        a = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        a[: n // 7] &= np.uint32(0x0fffffff)                       # uneven buckets
        make = lambda: clo.Sorter("satradix", ctx, "uint")         # noqa: E731
    elif kind == "ulong":
        a = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
        make = lambda: clo.Sorter("satradix", ctx, "ulong")        # noqa: E731
    elif kind == "narrow":
        a = (rng.integers(0, 1 << 20, n, dtype=np.uint32) | np.uint32(0x5a000000))
        make = lambda: clo.Sorter("satradix", ctx, "uint")         # noqa: E731
    else:   # few distinct keys: equal keys must keep their input order
        a = (rng.integers(0, 1000, n, dtype=np.uint64) << np.uint64(32 + 20)) | np.arange(n, dtype=np.uint64)
        make = lambda: clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)")   # noqa: E731
    qx = clo.Queue(ctx, profiling=True)
    qc = clo.Queue(ctx) if two_queues else None
    # (the switch is read when a sorter is made)
    monkeypatch.setenv("CLO_SORT_HOST_PIPELINE", "0")
    s0 = make()
    ref = s0.with_host_data(a, qx, qc)
    s0.close()
    monkeypatch.setenv("CLO_SORT_HOST_PIPELINE", "1")
    s = make()
    got = s.with_host_data(a, qx, qc)
    assert np.array_equal(got, ref)
    monkeypatch.delenv("CLO_SORT_HOST_PIPELINE")                    # the default: pipelined on a queue without profiling
    sd = make()
    qn = clo.Queue(ctx)
    assert np.array_equal(sd.with_host_data(a, qn, qc), ref)
    qn.close()
    sd.close()
    if kind != "pairs":
        assert np.array_equal(got, np.sort(a))
    else:
        assert np.array_equal(got, O.stable_sort(a, key_size=4, key_shift=32))
    again = s.with_host_data(a[: (1 << 24) + 1], qx, qc)           # the cached buffers serve a smaller array too
    assert np.array_equal(again, np.sort(a[: (1 << 24) + 1]) if kind != "pairs" else O.stable_sort(a[: (1 << 24) + 1], key_size=4, key_shift=32))
    for x in (s, qx, qc):
        if x is not None:
            x.close()


# ----------------------------------------------------------------------------
# gselect with compare / get_key outside the ahead-of-time family: compiled at run time
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("case", [("uint", "uint", "((x) ^ 0x5a5a5a5au)", None, lambda a: a ^ np.uint32(0x5a5a5a5a)),
                                  ("ulong", "uint", "(uint) (((x) >> 32) * 2654435761u)", None,
                                   lambda a: ((a >> np.uint64(32)) * np.uint64(2654435761)).astype(np.uint32)),
                                  ("int", "int", "(x)", "(((a) ^ 0x55) > ((b) ^ 0x55))", lambda a: a ^ np.int32(0x55))])
def test_gselect_with_runtime_compiled_macros(gpu, case):
    """Upstream pastes both macro bodies into its kernel (sort/clo_sort_gselect.cl:46-51) and builds
    it; here the same kernel text is compiled with hiprtc. Position = number of elements that
    compare before, ties (equal keys) by index: a stable sort by the order the macros define."""
    import cl_ops_amd as clo
    ctx, q = gpu
    et, kt, get_key, compare, model = case
    dt = clo.api.CLO_TYPE_NP[et]
    n = 5003
    rng = np.random.default_rng(11)
    if et == "ulong":
        a = (rng.integers(0, 300, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)   # ties: stability shows
    else:
        a = rng.integers(0, 1 << 20, n).astype(dt)
    s = clo.Sorter("gselect", ctx, et, key_type=kt, get_key=get_key, compare=compare)
    got = s.with_host_data(a, q)
    s.close()
    order = np.argsort(model(a), kind="stable")
    assert np.array_equal(got, a[order])


@pytest.mark.parametrize("key_bits", [28, 20, 12, 4])
@pytest.mark.parametrize("logn", [16, 20, 22])
def test_sweep_passes_with_an_empty_high_digit(gpu, monkeypatch, key_bits, logn):
    """A key width that is 4 modulo 8 makes the last single-sweep pass a one-digit pass (no second
    local split): it must still look back at ALL its predecessors (round 2's kernel took the rows it
    would have requested from inside the second split as 'valid, zero')."""
    import cl_ops_amd as clo
    ctx, q = gpu
    monkeypatch.setenv("CLO_RADIX_SWEEP", "1")
    n = (1 << logn) + 321
    a = O.bench_rand(key_bits + logn, "uint", n)
    mask = (1 << key_bits) - 1
    s = clo.Sorter("satradix", ctx, "uint", get_key="((x) & 0x%x)" % mask)
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, a[np.argsort(a & np.uint32(mask), kind="stable")])


# ----------------------------------------------------------------------------
# scans of every pair of types upstream's generic kernel accepts (clo_scan_abstract.c:122-125)
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("types", [("ulong", "uint"), ("uint", "uchar"), ("long", "short"), ("int", "ushort"), ("ulong", "int"),
                                   ("float", "uint"), ("double", "long"), ("float", "int"), ("half", "uint"), ("double", "uchar"),
                                   ("float", "ulong")])
@pytest.mark.parametrize("n", [1, 4097, (1 << 20) + 3, (1 << 24) + 5])
def test_scan_into_narrower_or_integer_sums(gpu, types, n):
    """Upstream's kernel converts every element with a C cast to the sum type and adds in that type
    (clo_scan_blelloch.cl:79-80): integer elements into a narrower sum keep their low bits, floating-point
    elements are truncated toward zero; the sums wrap. Bit-exact against numpy's astype + cumsum."""
    import cl_ops_amd as clo
    ctx, q = gpu
    et, st = types
    edt, sdt = clo.api.CLO_TYPE_NP[et], clo.api.CLO_TYPE_NP[st]
    rng = np.random.default_rng(n % 991 + len(et) + len(st))
    if np.issubdtype(edt, np.floating):
        si = np.iinfo(sdt)                                                 # (a float the sum type cannot hold is undefined in C: stay inside)
        hi = min(300.0, float(si.max))
        lo = -min(300.0, float(-si.min)) if si.min < 0 else 0.0
        a = (rng.random(n) * (hi - lo) * 0.999 + lo).astype(edt)
    else:
        info = np.iinfo(edt)
        a = rng.integers(info.min, info.max, n, dtype=edt, endpoint=True)
    sc = clo.Scanner("blelloch", ctx, et, st)
    got = sc.with_host_data(a, q)
    sc.close()
    cast = np.trunc(a.astype(np.float64)).astype(np.int64).astype(sdt) if np.issubdtype(edt, np.floating) else a.astype(sdt)
    wide = np.concatenate((np.zeros(1, np.uint64), np.cumsum(cast[:-1].astype(np.int64).astype(np.uint64), dtype=np.uint64)))
    exp = wide.astype(np.dtype("u%d" % sdt.itemsize)).view(sdt)
    assert got.dtype == sdt and np.array_equal(got, exp)


@pytest.mark.parametrize("et", ["half", "uchar", "float"])
def test_scan_with_half_sums(gpu, et):
    """A half sum type: every addition rounds to half precision (upstream: CLO_SCAN_SUM_TYPE half). Exact
    while every partial sum is a small integer, within a few half-ulps of the running magnitude otherwise,
    and the same bits from run to run."""
    import cl_ops_amd as clo
    ctx, q = gpu
    edt = clo.api.CLO_TYPE_NP[et]
    rng = np.random.default_rng(17)
    n = 40000
    a = (rng.random(n) < 0.04).astype(edt)                   # ~1600 ones: every prefix sum below 2048 is exact in half
    sc = clo.Scanner("blelloch", ctx, et, "half")
    got = sc.with_host_data(a, q)
    again = sc.with_host_data(a, q)
    assert got.dtype == np.float16 and np.array_equal(got.view(np.uint16), again.view(np.uint16))
    exact = np.concatenate(([0.0], np.cumsum(a.astype(np.float64))[:-1]))
    assert exact.max() < 2048 and np.array_equal(got.astype(np.float64), exact)
    if np.issubdtype(edt, np.floating):
        b = (rng.random(5000) * 0.5).astype(edt)
        got = sc.with_host_data(b, q).astype(np.float64)
        exact = np.concatenate(([0.0], np.cumsum(b.astype(np.float64))[:-1]))
        assert np.all(np.abs(got - exact) <= 64 * np.finfo(np.float16).eps * (exact + 1.0))
    sc.close()


# ----------------------------------------------------------------------------
# bitonic sorts of any numel with a key that is only part of the element
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("alg", ["sbitonic", "abitonic"])
@pytest.mark.parametrize("n", [2, 3, 1000, 4097, 100003])
@pytest.mark.parametrize("compare", [None, "((a) < (b))"])
def test_bitonic_any_numel_with_partial_keys(gpu, alg, n, compare):
    """Upstream's bitonic kernels have no bounds (powers of two only). Whole-element keys are padded with a
    sentinel; a key that is only part of the element cannot be (ties with the sentinel would show), so such
    sorts take the flip form of the network in place, comparators that reach past numel skipped: sorted by
    the key under `compare`, and a permutation of the input."""
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(n)
    keys = rng.integers(0, max(2, n // 3), n, dtype=np.uint64)             # many ties
    a = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    s = clo.Sorter(alg, ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)", compare=compare)
    got = s.with_host_data(a, q)
    s.close()
    k = (got >> np.uint64(32)).astype(np.int64)
    assert np.all(np.diff(k) >= 0) if compare is None else np.all(np.diff(k) <= 0)
    assert np.array_equal(np.sort(got), np.sort(a))


# ----------------------------------------------------------------------------
# a radix sort fed with its first digits by the producer of the keys
# ----------------------------------------------------------------------------

@pytest.mark.parametrize("kind,logn,shift,bits", [("uint", 26, 0, 32), ("ulong", 24, 32, 32), ("ulong", 24, 0, 64), ("uint", 20, 0, 32)])
def test_radix_sort_fed_with_first_digits(gpu, kind, logn, shift, bits):
    """clo_hip_radix_sort_fed: the caller hands over (elem >> key_shift) & 0xff per element and the first histogram
    reads those bytes instead of the elements (big tiles only; elsewhere they are ignored). Same result as the plain
    sort; the bytes may live in `tmp`."""
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    ctx, q = gpu
    dt = np.uint32 if kind == "uint" else np.uint64
    es = np.dtype(dt).itemsize
    n = (1 << logn) + 1234
    a = np.random.default_rng(logn + shift).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    key = (a >> dt(shift)) & dt((1 << bits) - 1 if bits < 64 else np.iinfo(dt).max)
    lib.clo_hip_env_refresh()        # (the switches are read when an object is made: none has been, in this test, since an earlier one changed them)
    takes = lib.clo_hip_radix_takes_first_digits(n, es, 0, 4)
    assert takes == (1 if n * es >= ((64 if es == 8 else 256) << 20) else 0)
    wsb = lib.clo_hip_radix_workspace_bytes(n, es, bits, 4)
    src, dst, tmp, ws = clo.Buffer(ctx, n * es), clo.Buffer(ctx, n * es), clo.Buffer(ctx, n * es), clo.Buffer(ctx, wsb)
    src.write(q, a)
    tmp.write(q, (key & dt(0xff)).astype(np.uint8))                        # the digits, in the scratch buffer
    ws.write(q, np.zeros(128, np.uint32))
    _hip.check(lib.clo_hip_radix_sort_fed(src.ptr, dst.ptr, tmp.ptr, n, es, shift, bits, 0, 4, tmp.ptr, ws.ptr, wsb, q.stream))
    q.finish()
    assert np.array_equal(dst.read(q, dt, n), a[np.argsort(key, kind="stable")])
    for b in (src, dst, tmp, ws):
        b.close()
