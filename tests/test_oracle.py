"""CPU tests of the oracle (oracle/clo_oracle.c): against the reference's own
known-answer checks (sortedness: clo_sort_bench.c:211-226; serial scan:
clo_scan_bench.c:252-271), against independent numpy implementations, against
the committed golden vectors, and of the structural contracts SURVEY.md §8a
names (digit-major counters, offsets semantics, launch counts)."""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sortscan_golden.npz"))
SIZES = [16, 64, 1024, 4096]


def test_bit_utils_follow_clo_common():
    L = O.lib()
    assert [L.clo_oracle_nlpo2(x) for x in (1, 2, 3, 5, 1000, 1024, 1025)] == [1, 2, 4, 8, 1024, 1024, 2048]
    assert [L.clo_oracle_tzc(1 << k) for k in range(0, 31)] == list(range(0, 31))
    assert L.clo_oracle_ones32(0xF0F0F0F0) == 16


def test_bench_rand_is_glib_mt19937():
    """g_rand_int stream == MT19937 init_genrand(seed); g_rand_double uses two
    draws (low word first). numpy's legacy RandomState has the same core."""
    r = np.random.RandomState(0)
    draws = r.randint(0, 2**32, size=8, dtype=np.uint64)
    exp = [np.uint32((float(draws[2 * i]) * 2.3283064365386963e-10 + float(draws[2 * i + 1]))
                     * 2.3283064365386963e-10 * 4294967295.0) for i in range(4)]
    assert list(O.bench_rand(0, "uint", 4)) == exp
    s = O.scan_bench_rand(0, np.uint32, 1000)
    assert s.min() >= 0 and s.max() < 128


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("name", ["u32_rand", "u32_all_equal", "u32_sorted", "u32_reverse", "u32_one_digit",
                                  "u32_two_digits", "u64_rand"])
def test_sorts_match_golden_and_reference_check(n, name):
    a, exp = GOLD["sort_%s_%d_in" % (name, n)], GOLD["sort_%s_%d_out" % (name, n)]
    for got in (O.sbitonic(a), O.abitonic(a)[0], O.abitonic(a, dev_max_lws=1024)[0],
                O.satradix(a, dev_max_lws=64), O.satradix(a, dev_max_lws=256) if n >= 256 else exp):
        assert O.check_sorted(got) == -1          # the reference's own pass criterion
        assert np.array_equal(got, exp)            # ... and the exact answer
        assert np.array_equal(np.sort(got), np.sort(a))  # permutation


@pytest.mark.parametrize("n", SIZES)
def test_pairs_stable_and_network_defined(n):
    a = GOLD["pairs_%d_in" % n]
    assert np.array_equal(O.satradix(a, key_size=4, key_shift=32, dev_max_lws=64), GOLD["pairs_%d_out" % n])
    assert np.array_equal(O.stable_sort(a, key_size=4, key_shift=32), GOLD["pairs_%d_out" % n])
    assert np.array_equal(O.sbitonic(a, key_size=4, key_shift=32), GOLD["pairs_%d_bitonic_out" % n])
    for kw in (dict(dev_max_lws=64), dict(dev_max_lws=256), dict(dev_max_lws=1024), dict(maxps=2), dict(minps=2),
               dict(maxsfs=6)):
        assert np.array_equal(O.abitonic(a, key_size=4, key_shift=32, **kw)[0], GOLD["pairs_%d_bitonic_out" % n]), kw


def test_satradix_reference_runs_redundant_passes_harmlessly():
    """ulong elements / uint key: upstream loops over 16 digits (element size)
    with OpenCL shift wrap; result must still be the stable sort by key."""
    rng = np.random.default_rng(0)
    k = rng.integers(0, 2**32, 2048, dtype=np.uint64)
    e = (k << np.uint64(32)) | np.arange(2048, dtype=np.uint64)
    assert np.array_equal(O.satradix(e, key_size=4, key_shift=32), e[np.argsort(k, kind="stable")])


@pytest.mark.parametrize("radix", [2, 4, 16, 256])
def test_satradix_radix_option(radix):
    a = O.bench_rand(3, "uint", 4096)
    assert np.array_equal(O.satradix(a, radix=radix, dev_max_lws=256), np.sort(a))


def test_satradix_radix_8_is_partial_upstream():
    """32/3 = 10 passes cover bits 0..29 only (clo_sort_satradix.c:168-169): the
    restatement keeps the quirk — sorted by the low 30 bits, stable."""
    a = O.bench_rand(4, "uint", 4096)
    got = O.satradix(a, radix=8, dev_max_lws=256)
    assert np.array_equal(got, a[np.argsort(a & 0x3FFFFFFF, kind="stable")])


def test_satradix_structural_fixture():
    a = GOLD["structural_in"]
    srt, offs, cnt, cs = O.satradix(a, radix=16, lws_max=64, dev_max_lws=64, debug=True)
    assert np.array_equal(offs, GOLD["structural_offsets"])
    assert np.array_equal(cnt, GOLD["structural_counters"])
    assert np.array_equal(cs, GOLD["structural_counters_sum"])
    assert np.array_equal(srt, GOLD["structural_out"])
    # offsets semantics (satradix.cl:152-201): first index of the digit in the
    # locally sorted tile; absent digit -> start of the next present one.
    L, R = 64, 16
    for w in range(1024 // L):
        d = np.sort(a[w * L:(w + 1) * L] & 15)
        for r in range(R):
            first = np.searchsorted(d, r, side="left")
            assert offs[w * R + r] == first


@pytest.mark.parametrize("n", SIZES)
def test_scans_match_golden(n):
    a = GOLD["scan_%d_in" % n]
    for sdt, tag in ((np.uint32, "u32"), (np.uint64, "u64")):
        exp = GOLD["scan_%s_%d_out" % (tag, n)]
        assert np.array_equal(O.serial_scan(a, sdt), exp)
        for lws in (8, 64, 256):
            if 2 * lws <= n:
                assert np.array_equal(O.blelloch(a, sdt, dev_max_lws=lws), exp)
    w = GOLD["scan_wrap_%d_in" % n]
    assert np.array_equal(O.blelloch(w, np.uint32, dev_max_lws=8), GOLD["scan_wrap_%d_out" % n])


def test_blelloch_serialises_blocks_and_skips_the_tail_like_upstream():
    # numel > 2*lws^2 -> several blocks per work-group (blelloch.c:135,140)
    a = O.scan_bench_rand(1, np.uint32, 1 << 14)
    assert np.array_equal(O.blelloch(a, np.uint64, dev_max_lws=16), O.serial_scan(a, np.uint64))
    # tail numel % (2*lws) is never scanned upstream (blelloch.cl:70); when the
    # third kernel runs it still adds the work-group sum there (blelloch.cl:209)
    b = O.scan_bench_rand(2, np.uint32, 96)
    got = O.blelloch(b, np.uint32, lws_max=32, dev_max_lws=32)
    assert np.array_equal(got[:64], O.serial_scan(b, np.uint32)[:64])
    assert not np.array_equal(got[64:], O.serial_scan(b, np.uint32)[64:])


def test_abitonic_launch_counts_match_survey():
    """62 launches at L=256 / 58 at L=1024 for 2^26. The count is
    a pure function of the strategy; check it on the schedule alone by sorting
    a small array with the same per-step kernels (T=16) and on the documented
    closed form for stages."""
    a = O.bench_rand(0, "uint", 1 << 16)
    _, l256 = O.abitonic(a, dev_max_lws=256)
    _, l1024 = O.abitonic(a, dev_max_lws=1024)
    assert (l256, l1024) == (22, 20)
    # sbitonic: T(T+1)/2 launches (clo_sort_sbitonic.c:102-118)
    assert 16 * 17 // 2 == 136


def test_mt_baseline_equals_serial():
    a = O.bench_rand(7, "uint", 1 << 15)
    assert np.array_equal(O.satradix(a, dev_max_lws=256, threads=4), O.satradix(a, dev_max_lws=256))
    s = O.scan_bench_rand(7, np.uint32, 1 << 15)
    assert np.array_equal(O.blelloch(s, np.uint32, dev_max_lws=64, threads=4), O.serial_scan(s, np.uint32))
    # the bitonic networks on several threads (the cpu_baseline of the sbitonic / abitonic legs): the same bits, ties included
    pairs = (np.random.default_rng(7).integers(0, 50, 1 << 13, dtype=np.uint64) << np.uint64(32)) | np.arange(1 << 13, dtype=np.uint64)
    for arr, kw in ((a, {}), (pairs, {"key_size": 4, "key_shift": 32})):
        assert np.array_equal(O.sbitonic(arr, threads=4, **kw), O.sbitonic(arr, **kw))
        for lws in (64, 256):
            mt, l_mt = O.abitonic(arr, dev_max_lws=lws, threads=4, **kw)
            st, l_st = O.abitonic(arr, dev_max_lws=lws, **kw)
            assert l_mt == l_st and np.array_equal(mt, st)


def test_typed_compare_and_descending():
    rng = np.random.default_rng(1)
    i = rng.integers(-1000, 1000, 1024).astype(np.int32)
    assert np.array_equal(O.sbitonic(i, key_kind=O.KEY_SIGNED), np.sort(i))
    f = ((rng.random(1024) - .5) * 100).astype(np.float32)
    assert np.array_equal(O.abitonic(f, key_kind=O.KEY_FLOAT)[0], np.sort(f))
    u = rng.integers(0, 2**32, 1024, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(O.sbitonic(u, descending=True), np.sort(u)[::-1])


# ---------------------------------------------------------------------------
# gselect restatement (clo_sort_gselect.cl:38-58): a stable rank sort
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 2, 17, 256, 1000])
def test_gselect_is_a_stable_sort_by_key(n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 20, n, dtype=np.uint64)
    e = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    got = O.gselect(e, key_size=4, key_shift=32)
    assert np.array_equal(got, O.stable_sort(e, key_size=4, key_shift=32))
    assert O.check_sorted(O.gselect(O.bench_rand(n, "uint", n))) == -1


def test_gselect_descending_and_signed():
    a = np.random.default_rng(0).integers(-1000, 1000, 500).astype(np.int32)
    assert np.array_equal(O.gselect(a, key_kind=O.KEY_SIGNED), np.sort(a))
    assert np.array_equal(O.gselect(a, key_kind=O.KEY_SIGNED, descending=True), np.sort(a)[::-1])


@pytest.mark.parametrize("n", [1, 17, 1000])
def test_gselect_matches_golden(n):
    a, exp = GOLD["gselect_pairs_%d_in" % n], GOLD["gselect_pairs_%d_out" % n]
    assert np.array_equal(O.gselect(a, key_size=4, key_shift=32), exp)
    assert np.array_equal(O.stable_sort(a, key_size=4, key_shift=32), exp)


@pytest.mark.parametrize("name,kind", [("i32", O.KEY_SIGNED), ("f32", O.KEY_FLOAT), ("f64", O.KEY_FLOAT)])
def test_typed_compare_matches_golden(name, kind):
    """The typed compare (clo_bench.c:26-65; CLO_SORT_COMPARE on the key type) in the
    oracle's bitonic and gselect restatements: numeric order, negatives included."""
    a, exp = GOLD["typed_%s_in" % name], GOLD["typed_%s_out" % name]
    assert np.array_equal(O.gselect(a, key_kind=kind), exp)
    assert np.array_equal(O.sbitonic(a, key_kind=kind), exp)
    assert np.array_equal(O.abitonic(a, key_kind=kind)[0], exp)
