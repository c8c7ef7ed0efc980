"""Seeded random sweeps over the sort / scan entry points (GPU): sizes that straddle
every path boundary (one-launch sorts, single-sweep passes, 8 192- and 16 384-element
tiles), every radix, element types, in place / out of place, stable pairs — against
numpy. Everything goes through the C-ABI of libcl_ops_hip.so."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

# boundaries of the radix paths, in elements: one launch <= 16384 (8192 of 8 bytes), sweeps up to
# 1024 tiles of 8192 (4096), big tiles from 64 MiB (8-byte) / 256 MiB (4-byte)
EDGES_4 = [1, 2, 17, 4095, 4096, 4097, 8192, 16384, 16385, 32768 + 1, (1 << 20) - 1, (1 << 23), (1 << 23) + 8193, (1 << 24) + 5]
EDGES_8 = [1, 3, 4096, 8192, 8193, 16385, (1 << 19) + 77, (1 << 22), (1 << 22) + 4097, (1 << 23) - 1, (1 << 23) + 16385]


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_satradix_keys(gpu, seed):
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(1000 + seed)
    for _ in range(8):
        et = rng.choice(["uchar", "ushort", "uint", "ulong", "int", "long", "float", "double"])
        dt = clo.api.CLO_TYPE_NP[et]
        edges = EDGES_8 if dt.itemsize == 8 else EDGES_4
        n = int(rng.choice(edges)) + int(rng.integers(0, 3))
        radix = int(rng.choice([2, 4, 8, 16, 32, 64, 128, 256]))
        if n < radix:
            n = radix
        if np.issubdtype(dt, np.floating):
            a = ((rng.random(n) - 0.5) * 10.0 ** int(rng.integers(0, 30))).astype(dt)
        else:
            info = np.iinfo(dt)
            span = int(rng.choice([8, 1 << 10, int(info.max) - int(info.min)]))   # few distinct values ... the whole range
            lo = int(info.min) if info.min < 0 and span > 1 << 10 else 0
            a = rng.integers(lo, lo + min(span, int(info.max) - lo), n, dtype=np.int64 if dt.itemsize < 8 or info.min < 0 else np.uint64, endpoint=True).astype(dt)
        s = clo.Sorter("satradix", ctx, et, options="radix=%d" % radix)
        exp = np.sort(a)
        if rng.integers(0, 2):
            got = s.with_host_data(a, q)
        else:   # device buffers, in place or into a second buffer
            src = clo.Buffer(ctx, a.nbytes)
            src.write(q, a)
            if rng.integers(0, 2):
                s.with_device_data(q, src, None, n)
                got = src.read(q, dt, n)
            else:
                dst = clo.Buffer(ctx, a.nbytes)
                s.with_device_data(q, src, dst, n)
                got = dst.read(q, dt, n)
                assert np.array_equal(src.read(q, dt, n).view(np.uint8), a.view(np.uint8)), "data_in changed"
                dst.close()
            src.close()
        s.close()
        assert np.array_equal(got.view(np.uint8), exp.view(np.uint8)), "type %s n %d radix %d" % (et, n, radix)


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_satradix_pairs_stable(gpu, seed):
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(2000 + seed)
    for _ in range(5):
        n = int(rng.choice(EDGES_8)) + int(rng.integers(0, 3))
        radix = int(rng.choice([4, 16, 64, 256]))
        n = max(n, radix)
        keys = rng.integers(0, int(rng.choice([4, 1 << 12, 1 << 32])), n, dtype=np.uint64)
        a = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        s = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint) ((x) >> 32)", options="radix=%d" % radix)
        got = s.with_host_data(a, q)
        s.close()
        assert np.array_equal(got, O.stable_sort(a, key_size=4, key_shift=32)), "n %d radix %d" % (n, radix)


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_scan(gpu, seed):
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(3000 + seed)
    one = None
    for _ in range(10):
        et, st = [("uint", "uint"), ("uint", "ulong"), ("uchar", "uint"), ("int", "long"), ("ushort", "ulong"), ("ulong", "ulong")][int(rng.integers(0, 6))]
        edt, sdt = clo.api.CLO_TYPE_NP[et], clo.api.CLO_TYPE_NP[st]
        n = int(rng.choice([1, 5, 4096, 16384, 16385, (1 << 20) + 3, (1 << 24) - 1, (1 << 24), (1 << 24) + 32769]))
        info = np.iinfo(edt)
        a = rng.integers(int(info.min), min(int(info.max), 1 << 40), n, dtype=np.int64, endpoint=False).astype(edt)
        sc = clo.Scanner("blelloch", ctx, et, st)
        got = sc.with_host_data(a, q)
        sc.close()
        wide = a.astype(np.int64 if info.min < 0 else np.uint64)
        exp = np.concatenate((np.zeros(1, wide.dtype), np.cumsum(wide[:-1], dtype=wide.dtype))).astype(sdt)
        assert np.array_equal(got, exp), "%s -> %s, n %d" % (et, st, n)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_key_fields(gpu, seed):
    """Random KEY FIELDS of the element (get_key = a shift and a mask of any width) at sizes on every path of the
    radix sort, radix 4 / 16 / 256: numpy's stable sort by the field. (Round 3: a 28-bit field on the single-sweep
    passes came out wrong with every test of whole-type keys green; tools/fuzz_keyfield_gpu.py is the long run.)"""
    import cl_ops_amd as clo
    ctx, q = gpu
    rng = np.random.default_rng(4000 + seed)
    sizes4 = [5, 4097, 16385, 40000, (1 << 17) + 3, (1 << 20) + 1, (1 << 22) + 77, (1 << 23) + 8193]
    sizes8 = [3, 8193, 30000, (1 << 16) + 3, (1 << 19) + 1, (1 << 21) + 77, (1 << 22) + 4097]
    for _ in range(10):
        es = int(rng.choice([4, 8]))
        et, dt = ("uint", np.uint32) if es == 4 else ("ulong", np.uint64)
        bits = 8 * es
        n = int(rng.choice(sizes4 if es == 4 else sizes8)) + int(rng.integers(0, 5))
        width = int(rng.integers(1, bits + 1))
        shift = int(rng.integers(0, bits - width + 1))
        mask = (1 << width) - 1
        a = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
        key = (a >> dt(shift)) & dt(mask)
        kt = "uint" if width <= 32 else "ulong"
        get_key = "(%s) (((x) >> %d) & 0x%x%s)" % (kt, shift, mask, "ul" if es == 8 else "u")
        radix = int(rng.choice([4, 16, 256]))
        s = clo.Sorter("satradix", ctx, et, key_type=kt, get_key=get_key, options="radix=%d" % radix)
        got = s.with_host_data(a, q)
        s.close()
        assert np.array_equal(got, a[np.argsort(key, kind="stable")]), "%s n=%d %s radix=%d" % (et, n, get_key, radix)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_segmented_sort(gpu, seed):
    """clo_hip_radix_sort_segmented (round 4): random segment counts and lengths (empty ones, one tile, many counter-scan
    chunks), the source as the segments back to back or as up to 256 pieces scattered over a larger array, key fields at
    any shift, both digit widths, both element sizes — against a stable numpy sort per segment."""
    import ctypes as C
    import torch
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib
    rng = np.random.default_rng(4000 + seed)
    for case in range(10):
        es = int(rng.choice([4, 8]))
        dt, tdt = (np.uint32, np.int32) if es == 4 else (np.uint64, np.int64)
        n = int(rng.choice([1, 37, 8192, 8193, 100000, (1 << 20) + 5, (1 << 22) + 12345]))
        nseg = int(rng.choice([1, 2, 7, 64, 255, 256]))
        cuts = np.sort(rng.integers(0, n + 1, nseg - 1)) if nseg > 1 else np.array([], dtype=np.int64)
        seg_counts = np.diff(np.concatenate(([0], cuts, [n]))).astype(np.int64)
        shift = int(rng.integers(0, 8 * es - 1))
        bits = int(rng.integers(1, 8 * es - shift + 1))
        digit_bits = int(rng.choice([4, 8]))
        pieces = bool(rng.integers(0, 2)) and nseg <= 64
        if pieces:          # every segment in up to 4 pieces, scattered over a source three times the size
            per = 4
            big = rng.integers(0, np.iinfo(dt).max, 3 * n + 5 * per * nseg + 64, dtype=dt, endpoint=True)
            pn, ps = [], []
            for k in range(nseg):
                c = np.sort(rng.integers(0, seg_counts[k] + 1, per - 1))
                for x in np.diff(np.concatenate(([0], c, [seg_counts[k]]))):
                    pn.append(int(x)); ps.append(k)
            order = rng.permutation(len(pn))                       # where the pieces lie has nothing to do with their order
            po = np.zeros(len(pn), dtype=np.int64)
            at = 3
            for i in order:
                po[i] = at
                at += pn[i] + int(rng.integers(0, 5))
            assert at <= big.size
            src_np = big
            gathered = np.concatenate([big[po[i]:po[i] + pn[i]] for i in range(len(pn))]) if n else big[:0]
        else:
            src_np = rng.integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
            gathered = src_np
        src = torch.from_numpy(src_np.view(tdt).copy()).cuda()
        ta = src if not pieces else torch.zeros(max(n, 1), dtype=src.dtype, device="cuda")
        tb = torch.zeros(max(n, 1), dtype=src.dtype, device="cuda")
        need = lib.clo_hip_radix_seg_workspace_bytes(n, nseg, es, digit_bits)
        ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
        in_b = C.c_int(-1)
        if pieces:
            npc = len(pn)
            extra = ((C.c_size_t * npc)(*pn), (C.c_size_t * npc)(*[int(x) for x in po]), (C.c_int * npc)(*ps), npc)
        else:
            extra = (None, None, None, 0)
        _hip.check(lib.clo_hip_radix_sort_segmented(src.data_ptr(), ta.data_ptr(), tb.data_ptr(), n, (C.c_size_t * nseg)(*[int(x) for x in seg_counts]), nseg,
                                                    *extra, es, shift, bits, digit_bits, ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream, C.byref(in_b)))
        torch.cuda.synchronize()
        got = (tb if in_b.value else ta).cpu().numpy().view(dt)[:n]
        exp = np.empty_like(gathered)
        at = 0
        mask = dt((1 << bits) - 1)
        for c in seg_counts:
            seg = gathered[at:at + c]
            exp[at:at + c] = seg[np.argsort((seg >> dt(shift)) & mask, kind="stable")]
            at += c
        assert np.array_equal(got, exp), "seed %d case %d: es %d n %d nseg %d shift %d bits %d digit %d pieces %s" % (seed, case, es, n, nseg, shift, bits, digit_bits, pieces)
        if pieces:
            assert np.array_equal(src.cpu().numpy().view(dt), src_np), "the source of a gathered sort changed"


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_sharded_sort_loopback(gpu, seed):
    """The sharded C path on one rank over real RCCL (`loopback=1`): random slice counts and key distributions that
    leave sub-buckets and whole slices EMPTY or put everything into one (fixed bits, a handful of values, one range of
    the key space, ascending keys, all equal), sizes on both sides of the slicing threshold — against numpy's sort.
    The same object sorts several inputs in a row (buffers, events and the adaptive table are reused)."""
    import torch
    from cl_ops_amd.multigpu import CShardedSorter
    rng = np.random.default_rng(5000 + seed)
    for etype in ("uint", "ulong"):
        dt, tdt = (np.uint32, np.int32) if etype == "uint" else (np.uint64, np.int64)
        top = int(np.iinfo(dt).max)
        opt = [None, "slices=1", "slices=2", "slices=4", "slices=8", "radix=256,slices=4"][int(rng.integers(0, 6))]
        s = CShardedSorter(etype, 0, options=(opt + "," if opt else "") + "loopback=1,slice_min=%d" % (16 << 20))
        for case in range(6):
            n = int(rng.choice([0, 1, 70000, (1 << 22) - 3, (1 << 22) + 4099, (1 << 23) + 17]))
            mode = int(rng.integers(0, 6))
            if mode == 0:
                a = rng.integers(0, top, n, dtype=dt, endpoint=True)
            elif mode == 1:     # some bits fixed
                a = (rng.integers(0, top, n, dtype=dt, endpoint=True) & dt(rng.integers(0, top, dtype=dt, endpoint=True) | rng.integers(0, top, dtype=dt, endpoint=True))) \
                    | dt(rng.integers(0, top, dtype=dt, endpoint=True) & rng.integers(0, top, dtype=dt, endpoint=True) & rng.integers(0, top, dtype=dt, endpoint=True))
            elif mode == 2:     # a handful of values
                vals = rng.integers(0, top, int(rng.integers(1, 17)), dtype=dt, endpoint=True)
                a = vals[rng.integers(0, vals.size, n)]
            elif mode == 3:     # one range of the key space
                lo = int(rng.integers(0, top, dtype=dt, endpoint=True))
                span = min(top - lo, top >> int(rng.integers(0, 8 * np.dtype(dt).itemsize - 4)))
                a = (dt(lo) + rng.integers(0, span, n, dtype=dt, endpoint=True)).astype(dt)
            elif mode == 4:     # ascending
                a = (np.arange(n, dtype=np.uint64) * np.uint64(max(1, (top // max(n, 1)) >> int(rng.integers(0, 8))))).astype(dt)
            else:               # all equal
                a = np.full(n, rng.integers(0, top, dtype=dt, endpoint=True), dtype=dt)
            t = torch.from_numpy(np.ascontiguousarray(a).view(tdt).copy()).cuda() if n else torch.empty(0, dtype=torch.int32 if etype == "uint" else torch.int64, device="cuda")
            out, m = s.sort(t)
            s.check()
            torch.cuda.synchronize()
            assert m == n, (etype, opt, mode, n)
            assert np.array_equal(out.cpu().numpy().view(dt)[:n], np.sort(a)), (etype, opt, mode, n)
        s.close()
