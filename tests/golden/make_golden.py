"""Generates tests/golden/*.npz — small input/expected-output vectors for the
sort/scan hot path (SURVEY.md §8c "Fixtures to commit").

The upstream tree holds no golden vectors for sort/scan and cannot be built or
run in this image, so these vectors are NOT reference outputs: expected values
are computed twice, by the CPU oracle (oracle/clo_oracle.c, which restates the
reference decomposition step by step) and by an independent numpy
implementation (stable argsort / cumsum); the script refuses to write a vector
on which the two disagree. The committed files pin both the oracle and the HIP
path against regressions.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SIZES = [16, 64, 1024, 4096]


def stable_by_key(e, key):
    return e[np.argsort(key, kind="stable")]


def main():
    out = {}
    for n in SIZES:
        # --- keys only, u32 / u64, reference bench distribution (clo_bench.c:103-119)
        u32 = O.bench_rand(n, "uint", n)
        u64 = O.bench_rand(n, "ulong", n)
        cases = {
            "u32_rand": u32,
            "u32_all_equal": np.full(n, 0xDEADBEEF, np.uint32),
            "u32_sorted": np.sort(u32),
            "u32_reverse": np.sort(u32)[::-1].copy(),
            # only one digit value present in every 4-bit digit (satradix.cl:165-201 gap filling)
            "u32_one_digit": np.full(n, 0x33333333, np.uint32),
            # two distinct digits, far apart, so most histogram offsets are back-filled
            "u32_two_digits": np.where(np.arange(n) % 3 == 0, 0x11111111, 0xEEEEEEEE).astype(np.uint32),
            "u64_rand": u64,
        }
        for name, a in cases.items():
            exp = np.sort(a)
            for alg, got in (("sbitonic", O.sbitonic(a)), ("abitonic", O.abitonic(a)[0]),
                             ("satradix", O.satradix(a, dev_max_lws=64))):
                assert np.array_equal(got, exp), (name, n, alg)
            out["sort_%s_%d_in" % (name, n)] = a
            out["sort_%s_%d_out" % (name, n)] = exp

        # --- (u32 key, u32 value) pairs, heavy duplication -> stability (BASELINE config 4)
        rng = np.random.default_rng(n)
        keys = rng.integers(0, 7, n, dtype=np.uint64)
        pairs = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        exp = stable_by_key(pairs, keys)
        got = O.satradix(pairs, key_size=4, key_shift=32, dev_max_lws=64)
        assert np.array_equal(got, exp), ("pairs", n)
        assert np.array_equal(O.stable_sort(pairs, key_size=4, key_shift=32), exp)
        out["pairs_%d_in" % n] = pairs
        out["pairs_%d_out" % n] = exp
        # bitonic on pairs: tie order is network-defined; keep the oracle's
        # network output (sbitonic and abitonic schedules must agree)
        sb = O.sbitonic(pairs, key_size=4, key_shift=32)
        ab, _ = O.abitonic(pairs, key_size=4, key_shift=32)
        assert np.array_equal(sb, ab), ("pairs bitonic", n)
        assert np.array_equal(sb >> np.uint64(32), np.sort(keys))
        assert np.array_equal(np.sort(sb), np.sort(pairs))
        out["pairs_%d_bitonic_out" % n] = sb

        # --- scans: bench distribution [0,128) (clo_scan_bench.c:219-223), and wrap-around
        s_in = O.scan_bench_rand(n, np.uint32, n)
        for sdt, tag in ((np.uint32, "u32"), (np.uint64, "u64")):
            exp = np.concatenate(([0], np.cumsum(s_in.astype(np.uint64))[:-1])).astype(sdt)
            assert np.array_equal(O.serial_scan(s_in, sdt), exp)
            assert np.array_equal(O.blelloch(s_in, sdt, dev_max_lws=min(64, max(n // 2, 1))), exp)
            out["scan_%s_%d_out" % (tag, n)] = exp
        out["scan_%d_in" % n] = s_in
        w_in = np.full(n, 0x90000000, np.uint32)
        exp = (np.arange(n, dtype=np.uint64) * np.uint64(0x90000000)).astype(np.uint32)
        assert np.array_equal(O.serial_scan(w_in, np.uint32), exp)
        out["scan_wrap_%d_in" % n] = w_in
        out["scan_wrap_%d_out" % n] = exp

    # --- gselect (clo_sort_gselect.cl:38-58): a stable rank sort, any numel
    for n in (1, 17, 1000):
        rng = np.random.default_rng(100 + n)
        keys = rng.integers(0, 5, n, dtype=np.uint64)
        pairs = (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        exp = stable_by_key(pairs, keys)
        assert np.array_equal(O.gselect(pairs, key_size=4, key_shift=32), exp), ("gselect", n)
        out["gselect_pairs_%d_in" % n] = pairs
        out["gselect_pairs_%d_out" % n] = exp

    # --- typed keys (signed / IEEE): numeric order, with negatives, both zeros and infinities.
    # Upstream's radix kernels order raw bits (its own typed check rejects that for negative
    # keys); the typed compare of its bitonic and gselect kernels gives this order. Expected =
    # numpy, cross-checked with the oracle's typed gselect.
    rng = np.random.default_rng(5)
    i32 = rng.integers(-2**31, 2**31 - 1, 1024, dtype=np.int64).astype(np.int32)
    f32 = (rng.standard_normal(1024) * 100).astype(np.float32)
    f32[:6] = [0.0, -0.0, np.inf, -np.inf, 1.5, -1.5]
    f64 = (rng.standard_normal(1024) * 1e6).astype(np.float64)
    for name, a, kind in (("i32", i32, O.KEY_SIGNED), ("f32", f32, O.KEY_FLOAT), ("f64", f64, O.KEY_FLOAT)):
        exp = np.sort(a, kind="stable")
        got = O.gselect(a, key_kind=kind)
        assert np.array_equal(got, exp), name           # numeric equality (-0 == +0)
        out["typed_%s_in" % name] = a
        out["typed_%s_out" % name] = exp

    # --- one digit pass of satradix, structural fixture (N=1024, L=64)
    a = O.bench_rand(1, "uint", 1024)
    srt, offs, cnt, cs = O.satradix(a, radix=16, lws_max=64, dev_max_lws=64, debug=True)
    # independent recomputation of the three aux arrays of the first pass
    L, R, W = 64, 16, 1024 // 64
    dig = (a & 15).reshape(W, L)
    cnt_np = np.stack([(dig == d).sum(axis=1) for d in range(R)]).astype(np.uint32).ravel()  # digit-major
    assert np.array_equal(cnt, cnt_np)
    assert np.array_equal(cs, np.concatenate(([0], np.cumsum(cnt_np)[:-1])).astype(np.uint32))
    assert np.array_equal(offs.reshape(W, R)[:, -1] + cnt.reshape(R, W)[-1], np.full(W, L, np.uint32))
    out["structural_in"] = a
    out["structural_offsets"] = offs
    out["structural_counters"] = cnt
    out["structural_counters_sum"] = cs
    out["structural_out"] = srt

    path = os.path.join(HERE, "sortscan_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
