"""Round-5 GPU tests: the segmented sort with pieces in two sources (clo_hip_radix_sort_segmented2: a rank's own share
of an exchange is gathered out of the partitioned shard and never copied), the bounded wait of the sharded sort's C API
on a healthy run, truthful local-memory introspection of the bitonic sorters, and the radix passes on tiles that end
exactly at the 64 KiB stage boundary (byte offsets kept in 16 bits since round 5)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cl_ops_amd  # noqa: F401
    torch.cuda.set_device(0)
    return torch


@pytest.mark.parametrize("dt,n,nseg,nsrc", [(np.uint32, (1 << 21) + 9, 8, 4), (np.uint64, (1 << 20) + 3, 32, 8),
                                            (np.uint32, (1 << 26) + 5, 8, 2), (np.uint32, 40000, 256, 1)])
def test_segmented_sort_with_a_second_source(gpu, dt, n, nseg, nsrc):
    """Every segment in `nsrc` pieces; the pieces of source 1 (source 0 when there is only one) lie in a SECOND array,
    at offsets of their own, the others in the first one with a hole where the absent pieces would have been."""
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    torch = gpu
    rng = np.random.default_rng(n % 977 + nsrc)
    own = min(1, nsrc - 1)
    sizes = rng.multinomial(n, np.ones(nseg * nsrc) / (nseg * nsrc)).reshape(nsrc, nseg)   # [source][segment]
    if nseg > 3:
        sizes[own, 3] += sizes[own, 1]
        sizes[own, 1] = 0                                    # an empty piece of the second source
    a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    off = np.concatenate(([0], np.cumsum(sizes.reshape(-1))[:-1])).reshape(nsrc, nseg)     # source-major layout of everything
    # second array: the own source's pieces back to back behind a prefix of junk; first array: the rest, the own pieces' room poisoned
    lead = 1237
    second = np.concatenate((rng.integers(0, 100, lead).astype(dt), a[off[own, 0]:off[own, 0] + sizes[own].sum()]))
    first = a.copy()
    first[off[own, 0]:off[own, 0] + sizes[own].sum()] = dt(0x5A5A5A5A)
    own_off = lead + np.concatenate(([0], np.cumsum(sizes[own])[:-1]))
    pn, po, ps, psrc = [], [], [], []
    for k in range(nseg):
        for p in range(nsrc):
            pn.append(int(sizes[p, k])); ps.append(k); psrc.append(int(p == own))
            po.append(int(own_off[k]) if p == own else int(off[p, k]))
    seg_counts = sizes.sum(axis=0)
    es = a.dtype.itemsize
    kb = 8 * es - 8
    tdt = np.int32 if es == 4 else np.int64
    t1 = torch.from_numpy(first.view(tdt)).cuda()
    t2 = torch.from_numpy(second.view(tdt)).cuda()
    tb = torch.full_like(t1, -1)
    need = lib.clo_hip_radix_seg_workspace_bytes(n, nseg, es, 4)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    npc = len(pn)
    in_b = C.c_int(-1)
    st = lib.clo_hip_radix_sort_segmented2(t1.data_ptr(), t2.data_ptr(), t1.data_ptr(), tb.data_ptr(), n, (C.c_size_t * nseg)(*[int(x) for x in seg_counts]), nseg,
                                           (C.c_size_t * npc)(*pn), (C.c_size_t * npc)(*po), (C.c_int * npc)(*ps), (C.c_int * npc)(*psrc), npc,
                                           es, 0, kb, 4, ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream, C.byref(in_b))
    _hip.check(st, "clo_hip_radix_sort_segmented2")
    torch.cuda.synchronize()
    got = (tb if in_b.value else t1).cpu().numpy().view(dt)
    exp = np.empty_like(a)
    at = 0
    mask = dt((1 << kb) - 1)
    for k in range(nseg):
        seg = np.concatenate([a[off[p, k]:off[p, k] + sizes[p, k]] for p in range(nsrc)])
        exp[at:at + seg.size] = seg[np.argsort(seg & mask, kind="stable")]
        at += seg.size
    assert np.array_equal(got, exp)
    assert np.array_equal(t2.cpu().numpy().view(dt), second)          # the second source is only read
    # a piece of the second source without one, or the second source being the first pass's target: refused
    st = lib.clo_hip_radix_sort_segmented2(t1.data_ptr(), None, t1.data_ptr(), tb.data_ptr(), n, (C.c_size_t * nseg)(*[int(x) for x in seg_counts]), nseg,
                                           (C.c_size_t * npc)(*pn), (C.c_size_t * npc)(*po), (C.c_int * npc)(*ps), (C.c_int * npc)(*psrc), npc,
                                           es, 0, kb, 4, ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream, C.byref(in_b))
    assert st == _hip.CLO_HIP_EARGS


def test_shard_sort_finish_on_a_healthy_run(gpu):
    """clo_shard_sort_finish with a bound: returns once the exchange and the sorts are done, aborts nothing; the result is
    right; a sort whose keys all lie in the own rank's bucket (here: always) moved no byte through the transport."""
    torch = gpu
    import cl_ops_amd as clo
    from cl_ops_amd.multigpu import CShardedSorter
    n = (1 << 22) + 12345
    keys = np.random.default_rng(11).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    t = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    ss = CShardedSorter("uint", 0, options="loopback=1,slices=4,slice_min=1,timeout_ms=20000")
    try:
        out, m = ss.sort(t)
        ss.finish(0)                     # the object's own bound
        ss.finish(5000)                  # and again (nothing left to wait for)
        got = out[:m].cpu().numpy().view(np.uint32)
        assert m == n and np.array_equal(got, np.sort(keys))
        x = ss.ss.exchange()
        assert x["slices"] == 4 and x["bytes_out"] == 0 and x["bytes_in"] == 0
    finally:
        ss.close()
    with pytest.raises(clo.CloError):
        CShardedSorter("uint", 0, options="loopback=1,timeout_ms=-5")


def test_bitonic_localmem_introspection_follows_numel(gpu):
    """clo_sort_get_localmem_usage of sbitonic / abitonic: the static LDS of the kernels the tiled schedule launches for
    `numel` — nothing below 32 elements (one launch per step), the run-time-schedule tile kernel up to one tile, the
    compile-time-schedule kernels (67 584 bytes) above."""
    import cl_ops_amd as clo
    ctx = clo.Context(0)
    for alg in ("sbitonic", "abitonic"):
        s = clo.Sorter(alg, ctx, "uint")
        idx = 0 if alg == "sbitonic" else 1      # abitonic: a "local" name (index 0 is abit_any: registers only)
        assert s.localmem_usage(idx, 0, 16) == 0
        assert s.localmem_usage(idx, 0, 1 << 10) == (8192 + 256) * 4
        assert s.localmem_usage(idx, 0, 1 << 14) == (16384 + 512) * 4
        assert s.localmem_usage(idx, 0, 1 << 26) == (16384 + 512) * 4
        s.close()
    s = clo.Sorter("abitonic", ctx, "ulong")
    assert s.localmem_usage(1, 0, 1 << 20) == (8192 + 256) * 8
    assert s.localmem_usage(0, 0, 1 << 20) == 0      # abit_any
    s.close()
    ctx.close()


@pytest.mark.parametrize("etype,dt", [("uint", np.uint32), ("ulong", np.uint64)])
@pytest.mark.parametrize("keys", ["no_top_digit", "all_low", "last_thread_full"])
def test_radix_tiles_that_end_at_the_64k_boundary(gpu, etype, dt, keys):
    """A full 64 KiB tile (16 384 uint32 / 8 192 uint64) whose highest digits are absent: the ends of the present digits
    sit at the very end of the stage, byte offset 65 536 = 0 mod 2^16 — the case the 16-bit byte offsets wrap on."""
    import cl_ops_amd as clo
    n = 1 << 26 if dt == np.uint32 else 1 << 23        # big tiles (>= 256 MiB / 32 MiB of 8-byte elements): the 64 KiB stage
    rng = np.random.default_rng(3)
    a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    if keys == "no_top_digit":
        a &= dt(0x7777777777777777 & np.iinfo(dt).max)       # no digit above 7 anywhere: every digit 15 .. 8 is absent in every pass
    elif keys == "all_low":
        a &= dt(0x0101010101010101 & np.iinfo(dt).max)       # digits 0 and 1 only
    else:
        a[::2] = a[::2] & dt(0x0f0f0f0f0f0f0f0f & np.iinfo(dt).max)   # a mix: half of the keys with empty high nibbles
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    s = clo.Sorter("satradix", ctx, etype)
    src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
    src.write(q, a)
    s.with_device_data(q, src, dst, n)
    got = dst.read(q, dt, n)
    assert np.array_equal(got, np.sort(a))
    for b in (src, dst):
        b.close()
    s.close()
    q.close()
    ctx.close()


@pytest.mark.parametrize("etype,dt,log2n", [("uint", np.uint32, 20), ("int", np.int32, 21), ("float", np.float32, 20),
                                            ("ulong", np.uint64, 19), ("long", np.int64, 20), ("uint", np.uint32, 24)])
def test_abitonic_two_tile_merge_modes(gpu, etype, dt, log2n):
    """The tiled bitonic schedule with no two-tile merge pass (CLO_BITONIC_MERGE2=0), with one where it saves a strided
    pass (the default) and with one at EVERY stage above the tile (2): the same network cut into launches three ways —
    the same result, ascending and descending, also for a size that is no power of two."""
    import os
    import cl_ops_amd as clo
    from cl_ops_amd._hip import lib
    rng = np.random.default_rng(log2n)
    n = (1 << log2n) - (12345 if log2n == 21 else 0)
    if np.issubdtype(dt, np.floating):
        a = rng.standard_normal(n).astype(dt)
    else:
        a = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    old = os.environ.get("CLO_BITONIC_MERGE2")
    try:
        for mode in ("0", "1", "2"):
            os.environ["CLO_BITONIC_MERGE2"] = mode
            lib.clo_hip_env_refresh()
            for opts, exp in ((None, np.sort(a)), ("desc", np.sort(a)[::-1])):
                s = clo.Sorter("abitonic", ctx, etype, compare="((a) < (b))" if opts else None)   # (upstream's compare says when to SWAP: "<" sorts descending)
                src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
                src.write(q, a)
                s.with_device_data(q, src, dst, n)
                got = dst.read(q, dt, n)
                assert np.array_equal(got, exp), (mode, opts)
                for b in (src, dst):
                    b.close()
                s.close()
    finally:
        if old is None:
            os.environ.pop("CLO_BITONIC_MERGE2", None)
        else:
            os.environ["CLO_BITONIC_MERGE2"] = old
        lib.clo_hip_env_refresh()
        q.close()
        ctx.close()
