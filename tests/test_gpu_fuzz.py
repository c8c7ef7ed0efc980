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
