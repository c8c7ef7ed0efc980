"""Developer probe: A/B timing of library builds on one box, interleaved.
  python tools/ab_probe.py --libs lib,lib_old --jobs sort:uint:28,scan:uint:26 [--rounds 2] [--steps 20]
Every (library, round) is a child process with CLO_HIP_LIBRARY set; inside it every job is timed the way bench.py
times a step: K back-to-back calls between one pair of events (plus the per-kernel-family event times of K more calls).
Jobs: sort:<uint|ulong|pairs>:<log2 n>[:radix]   scan:<uint|ulong>:<log2 n>   abitonic:<log2 n>[:uint|ulong|float]   sbitonic:<log2 n>"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def child(jobs, steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import numpy as np
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib

    ctx = clo.Context(0)
    q = clo.Queue(ctx)

    def run(fn, labels):
        fn()
        q.finish()
        t = clo.HipEventTimer(q)
        best = 1e9
        for _ in range(3):
            t.start()
            for _ in range(steps):
                fn()
            t.stop()
            best = min(best, t.elapsed_ms() / steps)
        t.close()
        lib.clo_hip_timing_enable(1)
        lib.clo_hip_timing_reset()
        for _ in range(steps):
            fn()
        q.finish()
        k = {l: _hip.timing_read(l) for l in labels}
        lib.clo_hip_timing_enable(0)
        lib.clo_hip_timing_reset()
        return best, " ".join("%s=%.4f(x%d)" % (l, k[l][1] / max(k[l][0], 1), k[l][0] // steps) for l in labels if k[l][0])

    for job in jobs:
        f = job.split(":")
        rng = np.random.default_rng(0)
        if f[0] == "sort":
            kind, logn = f[1], int(f[2])
            radix = int(f[3]) if len(f) > 3 else 16
            n = 1 << logn
            if kind == "uint":
                a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                s = clo.Sorter("satradix", ctx, "uint", options="radix=%d" % radix)
            elif kind == "pairs":
                a = (rng.integers(0, 2**32, n, dtype=np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
                s = clo.Sorter("satradix", ctx, "ulong", key_type="uint", get_key="(uint)((x)>>32)", options="radix=%d" % radix)
            else:
                a = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
                s = clo.Sorter("satradix", ctx, "ulong", options="radix=%d" % radix)
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
            src.write(q, a)
            ms, k = run(lambda: s.with_device_data(q, src, dst, n), ["radix_hist", "radix_offsets", "radix_pass", "radix_sweep", "radix_ghist", "radix_small"])
            got = dst.read(q, a.dtype, n)
            key = got if kind != "pairs" else got >> np.uint64(32)
            ok = bool(np.all(key[:-1] <= key[1:])) and int(np.bitwise_xor.reduce(got)) == int(np.bitwise_xor.reduce(a))
            print("RESULT %s ms=%.4f Mkeys/s=%.0f ok=%s | %s" % (job, ms, n / ms / 1e3, ok, k), flush=True)
            src.close(); dst.close(); s.close()
        elif f[0] == "scan":
            st, logn = f[1], int(f[2])
            n = 1 << logn
            a = rng.integers(0, 128, n).astype(np.uint32)
            sc = clo.Scanner("blelloch", ctx, "uint", st)
            sdt = clo.api.CLO_TYPE_NP[st]
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, n * sdt.itemsize)
            src.write(q, a)
            ms, k = run(lambda: sc.with_device_data(q, src, dst, n), ["scan"])
            got = dst.read(q, sdt, n)
            exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))[:-1])).astype(sdt)
            print("RESULT %s ms=%.4f MValues/s=%.0f TB/s=%.2f ok=%s | %s" % (job, ms, n / ms / 1e3, n * (4 + sdt.itemsize) / ms / 1e9, bool(np.array_equal(got, exp)), k), flush=True)
            src.close(); dst.close(); sc.close()
        elif f[0] in ("abitonic", "sbitonic"):
            n = 1 << int(f[1])
            et = f[2] if len(f) > 2 else "uint"   # abitonic:<log2 n>[:uint|ulong|float]
            bdt = {"uint": np.uint32, "ulong": np.uint64, "float": np.float32}[et]
            a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(bdt) if et != "ulong" else rng.integers(0, 2**63, n, dtype=np.uint64)
            s = clo.Sorter(f[0], ctx, et)
            src, dst = clo.Buffer(ctx, a.nbytes), clo.Buffer(ctx, a.nbytes)
            src.write(q, a)
            ms, k = run(lambda: s.with_device_data(q, src, dst, n), ["bitonic_presort", "bitonic_tile", "bitonic_strided", "bitonic_strided2", "bitonic_step"])
            got = dst.read(q, bdt, n)
            print("RESULT %s ms=%.4f Mkeys/s=%.0f ok=%s | %s" % (job, ms, n / ms / 1e3, bool(np.all(got[:-1] <= got[1:])), k), flush=True)
            src.close(); dst.close(); s.close()
    q.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="lib")
    ap.add_argument("--jobs", default="sort:uint:28,scan:uint:26")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    jobs = a.jobs.split(",")
    if a.child:
        child(jobs, a.steps)
        return
    for r in range(a.rounds):
        for name in a.libs.split(","):
            path = os.path.join(ROOT, "cl_ops_amd", name, "libcl_ops_hip.so")
            env = dict(os.environ, CLO_HIP_LIBRARY=path)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--jobs", a.jobs, "--steps", str(a.steps)],
                env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            for line in out.stdout.splitlines():
                if line.startswith("RESULT"):
                    print("[%s r%d] %s" % (name, r, line[7:]), flush=True)
            if out.returncode != 0:
                print("[%s r%d] child failed (%d):\n%s" % (name, r, out.returncode, out.stdout[-2000:]), flush=True)


if __name__ == "__main__":
    main()
