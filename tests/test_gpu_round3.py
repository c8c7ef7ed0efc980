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
# give-ups of the single-sweep radix passes (the library's path for 2^15 .. 2^23 keys)
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
    monkeypatch.setenv("CLO_R1_POOLS", pools)
    n = (1 << logn) + 12345
    a = O.bench_rand(logn, kind, n)
    s = clo.Sorter("satradix", ctx, kind)
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, np.sort(a))


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
