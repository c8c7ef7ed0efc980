"""Round-4 GPU parity tests: the segmented sort (clo_hip_radix_sort_segmented, the local step of the sharded sort
since round 4) and the 8-bit MSD partition, through the C-ABI, bit-exact against numpy."""
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


def _seg_sort(torch, a, seg_counts, key_shift, key_bits, digit_bits=4, pieces=None):
    """Runs clo_hip_radix_sort_segmented on the array `a` (numpy, uint32 / uint64); returns the result (numpy)."""
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    es = a.dtype.itemsize
    n = a.size
    tdt = np.int32 if es == 4 else np.int64
    ta = torch.from_numpy(a.view(tdt).copy()).cuda()
    tb = torch.full_like(ta, -1)
    nseg = len(seg_counts)
    need = lib.clo_hip_radix_seg_workspace_bytes(n, nseg, es, digit_bits)
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    sc = (C.c_size_t * nseg)(*[int(x) for x in seg_counts])
    if pieces is not None:
        pn, po, ps = pieces
        npc = len(pn)
        args = ((C.c_size_t * npc)(*[int(x) for x in pn]), (C.c_size_t * npc)(*[int(x) for x in po]), (C.c_int * npc)(*[int(x) for x in ps]), npc)
    else:
        args = (None, None, None, 0)
    in_b = C.c_int(-1)
    st = lib.clo_hip_radix_sort_segmented(ta.data_ptr(), ta.data_ptr(), tb.data_ptr(), n, sc, nseg, *args, es, key_shift, key_bits, digit_bits,
                                          ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream, C.byref(in_b))
    _hip.check(st, "clo_hip_radix_sort_segmented")
    torch.cuda.synchronize()
    assert in_b.value == (-(-key_bits // 8)) % 2
    return (tb if in_b.value else ta).cpu().numpy().view(a.dtype)


def _expect(a, seg_counts, key_shift, key_bits):
    out = np.empty_like(a)
    at = 0
    mask = a.dtype.type((1 << key_bits) - 1)
    for c in seg_counts:
        seg = a[at:at + c]
        k = (seg >> a.dtype.type(key_shift)) & mask
        out[at:at + c] = seg[np.argsort(k, kind="stable")]
        at += c
    return out


@pytest.mark.parametrize("dt,n,nseg,key_bits", [
    (np.uint32, 1 << 20, 16, 24), (np.uint32, (1 << 22) + 12345, 64, 24), (np.uint32, 70001, 5, 24),
    (np.uint64, (1 << 20) + 77, 32, 56), (np.uint32, 1 << 18, 256, 24), (np.uint32, (1 << 21) + 3, 7, 20),
    (np.uint32, 1 << 16, 3, 8), (np.uint64, 1 << 19, 9, 16), (np.uint32, 5000, 256, 24)])
def test_segmented_sort_contiguous(gpu, dt, n, nseg, key_bits):
    rng = np.random.default_rng(n + nseg)
    a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    cuts = np.sort(rng.integers(0, n + 1, nseg - 1))
    if nseg >= 5:
        cuts[1] = cuts[0]                         # an empty segment
    seg_counts = np.diff(np.concatenate(([0], cuts, [n])))
    got = _seg_sort(gpu, a, seg_counts, 0, key_bits)
    assert np.array_equal(got, _expect(a, seg_counts, 0, key_bits))


@pytest.mark.parametrize("dt,logn", [(np.uint32, 26), (np.uint64, 24)])
def test_segmented_sort_big_tiles_and_digit_stream(gpu, dt, logn):
    """>= 256 MiB (32 MiB of 8-byte elements): 16 384-element tiles, the digit bytes between the passes."""
    n = (1 << logn) + 4099
    rng = np.random.default_rng(logn)
    a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    nseg = 64
    seg_counts = np.full(nseg, n // nseg)
    seg_counts[-1] += n - seg_counts.sum()
    seg_counts[3] += 1000
    seg_counts[4] -= 1000
    kb = 8 * a.dtype.itemsize - 8
    got = _seg_sort(gpu, a, seg_counts, 0, kb)
    assert np.array_equal(got, _expect(a, seg_counts, 0, kb))


def test_segmented_sort_one_huge_segment_among_small_ones(gpu):
    """Skew: one segment of many counter-scan chunks between tiny ones."""
    n = (1 << 23) + 17
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    seg_counts = np.array([3, 0, 100, n - 3 - 100 - 9000 - 1, 9000, 1])
    got = _seg_sort(gpu, a, seg_counts, 0, 24)
    assert np.array_equal(got, _expect(a, seg_counts, 0, 24))


@pytest.mark.parametrize("dt,n,nseg,nsrc", [(np.uint32, (1 << 21) + 9, 8, 4), (np.uint64, 1 << 20, 32, 8), (np.uint32, (1 << 26) + 5, 8, 8)])
def test_segmented_sort_gathers_pieces(gpu, dt, n, nseg, nsrc):
    """The source holds every segment in `nsrc` pieces, laid out source-major (what a rank of the sharded sort
    receives: for every source rank its sub-buckets back to back); the result holds the segments back to back."""
    rng = np.random.default_rng(n % 1000 + nsrc)
    sizes = rng.multinomial(n, np.ones(nseg * nsrc) / (nseg * nsrc)).reshape(nsrc, nseg)   # [source][segment]
    sizes[1, 2] += sizes[0, 2]
    sizes[0, 2] = 0                                   # an empty piece
    a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    off = np.concatenate(([0], np.cumsum(sizes.reshape(-1))[:-1])).reshape(nsrc, nseg)     # source-major layout
    pn, po, ps = [], [], []
    for k in range(nseg):
        for p in range(nsrc):
            pn.append(sizes[p, k]); po.append(off[p, k]); ps.append(k)
    seg_counts = sizes.sum(axis=0)
    kb = 8 * a.dtype.itemsize - 8
    got = _seg_sort(gpu, a, seg_counts, 0, kb, pieces=(pn, po, ps))
    # expected: segment k = its pieces in source order, stably sorted by the low bits
    gathered = np.concatenate([a[off[p, k]:off[p, k] + sizes[p, k]] for k in range(nseg) for p in range(nsrc)])
    assert np.array_equal(got, _expect(gathered, seg_counts, 0, kb))


@pytest.mark.parametrize("dt,bits", [(np.uint32, 7), (np.uint32, 8), (np.uint64, 8)])
def test_msd_partition_on_seven_and_eight_bits(gpu, dt, bits):
    torch = gpu
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    n = (1 << 22) + 333
    es = np.dtype(dt).itemsize
    a = np.random.default_rng(bits).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    tdt = np.int32 if es == 4 else np.int64
    src = torch.from_numpy(a.view(tdt).copy()).cuda()
    dst = torch.empty_like(src)
    counts = torch.zeros(1 << bits, dtype=torch.int64, device="cuda")
    need = lib.clo_hip_msd_workspace_bytes(n, es, bits)
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    _hip.check(lib.clo_hip_msd_partition(src.data_ptr(), dst.data_ptr(), n, es, 0, 8 * es, bits, counts.data_ptr(), ws.data_ptr(), need,
                                         torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    top = a >> dt(8 * es - bits)
    assert np.array_equal(dst.cpu().numpy().view(dt), a[np.argsort(top, kind="stable")])
    assert np.array_equal(counts.cpu().numpy(), np.bincount(top.astype(np.int64), minlength=1 << bits))


# ----------------------------------------------------------------------------
# compiler_opts reach the run-time compiler (sort/clo_sort_abstract.c:173-179)
# ----------------------------------------------------------------------------

@pytest.fixture(scope="module")
def cq(gpu):
    import cl_ops_amd as clo
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    yield ctx, q
    q.close()
    ctx.close()


@pytest.mark.parametrize("alg", ["sbitonic", "abitonic", "gselect", "satradix"])
@pytest.mark.parametrize("opts", ["-DSHIFT=12", "-cl-fast-relaxed-math -D SHIFT=12 -DUNUSED"])
def test_compiler_opts_define_a_macro_used_in_get_key(cq, alg, opts):
    """Upstream hands compiler_opts to the OpenCL JIT with the kernel source, so a caller may define SHIFT there and
    use it inside get_key; here the options reach hiprtc for all four sorters."""
    import cl_ops_amd as clo
    ctx, q = cq
    n = 3000 if alg == "gselect" else (1 << 15) + (0 if alg != "satradix" else 77)
    a = np.random.default_rng(12).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter(alg, ctx, "uint", get_key="((x) >> SHIFT) & 0xfff", compiler_opts=opts)
    got = s.with_host_data(a, q)
    s.close()
    key = (a >> np.uint32(12)) & np.uint32(0xfff)
    if alg in ("gselect", "satradix"):            # stable sorters: the result is unique
        assert np.array_equal(got, a[np.argsort(key, kind="stable")])
    else:                                          # bitonic networks: sorted by the key, the same multiset
        gk = (got >> np.uint32(12)) & np.uint32(0xfff)
        assert np.all(gk[:-1] <= gk[1:]) and np.array_equal(np.sort(got), np.sort(a))


def test_compiler_opts_also_change_a_parseable_expression(cq):
    """`(x) >> S` would be refused by the parser (S is not a number) — but `(x) & MASK`-style names that the options
    define must never be PARSED as something else either: the define decides."""
    import cl_ops_amd as clo
    ctx, q = cq
    a = np.random.default_rng(13).integers(0, 1 << 32, 1 << 16, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter("satradix", ctx, "uint", get_key="(x) >> S", compiler_opts="-DS=20")
    got = s.with_host_data(a, q)
    s.close()
    assert np.array_equal(got, a[np.argsort(a >> np.uint32(20), kind="stable")])


@pytest.mark.parametrize("alg", ["abitonic", "satradix"])
@pytest.mark.parametrize("opts,needle", [("-DSHIFT=(", "error"), ("-fno-such-flag-at-all -DSHIFT=3", "no-such-flag")])
def test_bad_compiler_opts_give_a_gerror_with_the_build_log(cq, alg, opts, needle):
    import cl_ops_amd as clo
    ctx, q = cq
    with pytest.raises(clo.CloError) as e:
        clo.Sorter(alg, ctx, "uint", get_key="((x) >> SHIFT) & 0xfff", compiler_opts=opts)
    assert e.value.code == 2                                   # CLO_ERROR_ARGS (clo_common.in.h:80-95)
    assert "Could not build kernels" in e.value.message and needle in e.value.message.lower(), e.value.message


@pytest.mark.parametrize("dt,key_bits", [(np.uint32, 24), (np.uint32, 16), (np.uint64, 24)])
def test_segmented_sort_from_a_larger_source(gpu, dt, key_bits):
    """`src` is neither of the two working buffers: the pieces lie anywhere in a larger array (what the pipelined
    clo_sort_with_host_data hands over: sub-buckets scattered over the split chunks), `src` stays untouched, and the
    result lands in `b` after an odd number of passes, in `a` after an even one."""
    torch = gpu
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    es = np.dtype(dt).itemsize
    rng = np.random.default_rng(key_bits + es)
    big = rng.integers(0, np.iinfo(dt).max, (1 << 22) + 333, dtype=dt, endpoint=True)
    nseg, nchunk = 16, 8
    sizes = rng.integers(0, 9000, (nseg, nchunk))
    starts = np.sort(rng.choice(big.size // 16384, nseg * nchunk, replace=False)).reshape(nchunk, nseg).T * 16384 + 5   # disjoint, unaligned
    pn, po, ps = [], [], []
    for k in range(nseg):
        for c in range(nchunk):
            pn.append(int(sizes[k, c])); po.append(int(starts[k, c])); ps.append(k)
    seg_counts = sizes.sum(axis=1)
    n = int(seg_counts.sum())
    tdt = np.int32 if es == 4 else np.int64
    src = torch.from_numpy(big.view(tdt).copy()).cuda()
    ta = torch.zeros(n, dtype=src.dtype, device="cuda")
    tb = torch.zeros(n, dtype=src.dtype, device="cuda")
    need = lib.clo_hip_radix_seg_workspace_bytes(n, nseg, es, 8)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    npc = len(pn)
    in_b = C.c_int(-1)
    _hip.check(lib.clo_hip_radix_sort_segmented(src.data_ptr(), ta.data_ptr(), tb.data_ptr(), n, (C.c_size_t * nseg)(*[int(x) for x in seg_counts]), nseg,
                                                (C.c_size_t * npc)(*pn), (C.c_size_t * npc)(*po), (C.c_int * npc)(*ps), npc, es, 3, key_bits, 8,
                                                ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream, C.byref(in_b)))
    torch.cuda.synchronize()
    assert in_b.value == (-(-key_bits // 8)) % 2
    got = (tb if in_b.value else ta).cpu().numpy().view(dt)
    gathered = np.concatenate([big[starts[k, c]:starts[k, c] + sizes[k, c]] for k in range(nseg) for c in range(nchunk)])
    assert np.array_equal(got, _expect(gathered, seg_counts, 3, key_bits))
    assert np.array_equal(src.cpu().numpy().view(dt), big)


def test_sweep_pass_clock_stamps_diagnostic(gpu, monkeypatch):
    """clo_hip_radix_debug_stamps (include/clo_hip.h): while a buffer is registered, the last single-sweep pass of a sort
    leaves eight words per tile — seven s_memtime stamps in program order, then (XCC id << 32 | work-group) — and
    nothing is written once the buffer is taken away again. The sort's result is the same either way."""
    torch = gpu
    import cl_ops_amd as clo
    from cl_ops_amd._hip import lib
    lib.clo_hip_radix_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
    monkeypatch.setenv("CLO_RADIX_SWEEP", "1")
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    n = (1 << 20) + 77
    a = np.random.default_rng(3).integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    s = clo.Sorter("satradix", ctx, "uint")               # (reads the switch)
    tiles_max = 4096
    stamps = torch.zeros(tiles_max * 8, dtype=torch.int64, device="cuda")
    assert lib.clo_hip_radix_debug_stamps(stamps.data_ptr(), tiles_max) == 0
    try:
        got = s.with_host_data(a, q)
    finally:
        assert lib.clo_hip_radix_debug_stamps(None, 0) == 0
    assert np.array_equal(got, np.sort(a))
    st = stamps.cpu().numpy().view(np.uint64).reshape(tiles_max, 8)
    used = np.flatnonzero(st[:, 0])
    assert used.size >= n // 16384 and used.size == used[-1] + 1     # tiles 0 .. T-1 of the pass, whatever its tile shape
    t = st[used, :7].astype(np.int64)
    assert np.all(np.diff(t, axis=1) >= 0)                            # stamps in program order
    assert np.all((st[used, 7] >> np.uint64(32)) < 8)                 # an XCC id
    stamps.zero_()
    again = s.with_host_data(a, q)
    torch.cuda.synchronize()
    assert np.array_equal(again, got) and int(stamps.abs().sum().item()) == 0
    s.close()
    q.close()
    ctx.close()
