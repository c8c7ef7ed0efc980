"""Developer soak (GPU box): a handful of long-lived sorter / scanner objects, used over and over on arrays whose sizes
jump around (their cached workspaces grow, are reused for smaller arrays, see other tile shapes and pass kinds), every
result checked against numpy. What the fuzz tools do not cover: state a call leaves behind for the next one (tickets,
epochs, hand-off words, cached buffers). python tools/soak_gpu.py [seconds=240] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cl_ops_amd as clo  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = clo.Context(0)
    q = clo.Queue(ctx)
    objs = [
        ("satradix", "uint", clo.Sorter("satradix", ctx, "uint")),
        ("satradix", "ulong", clo.Sorter("satradix", ctx, "ulong")),
        ("satradix256", "uint", clo.Sorter("satradix", ctx, "uint", options="radix=256")),
        ("satradix", "ushort", clo.Sorter("satradix", ctx, "ushort")),
        ("abitonic", "uint", clo.Sorter("abitonic", ctx, "uint")),
        ("sbitonic", "int", clo.Sorter("sbitonic", ctx, "int")),
        ("scan", "uint", clo.Scanner("blelloch", ctx, "uint", "uint")),
        ("scan64", "uint", clo.Scanner("blelloch", ctx, "uint", "ulong")),
    ]
    import torch
    from cl_ops_amd.multigpu import CShardedSorter
    objs += [("shard", "uint", CShardedSorter("uint", 0, options="loopback=1,slice_min=%d" % (16 << 20))), ("shard", "ulong", CShardedSorter("ulong", 0, options="loopback=1,slices=4,slice_min=%d" % (32 << 20))),
             ("hostsort", "uint", objs[0][2]), ("hostsort", "ulong", objs[1][2])]
    cap = 1 << 25
    src, dst = clo.Buffer(ctx, cap * 8), clo.Buffer(ctx, cap * 8)
    t0 = time.time()
    calls = bad = 0
    last_print = t0
    while time.time() - t0 < seconds:
        name, et, o = objs[int(rng.integers(0, len(objs)))]
        dt = clo.api.CLO_TYPE_NP[et]
        top = 25 if name.startswith("satradix") or name.startswith("scan") else (22 if name == "abitonic" else 16)
        logn = int(rng.integers(1, top + 1))
        n = int(rng.integers(max(1, 1 << (logn - 1)), (1 << logn) + 1))
        if name == "shard":        # the sharded C path over RCCL on this one rank: sizes on both sides of the slicing threshold
            n = int(rng.integers(0, 1 << int(rng.integers(10, 25))))
            mode = int(rng.integers(0, 3))
            top_v = int(np.iinfo(dt).max)
            a = rng.integers(0, top_v, n, dtype=dt, endpoint=True) if mode == 0 else (rng.integers(0, top_v, n, dtype=dt, endpoint=True) >> dt.type(int(rng.integers(1, 12)))) if mode == 1 \
                else np.full(n, rng.integers(0, top_v, dtype=dt, endpoint=True), dtype=dt)
            t = torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if et == "uint" else np.int64).copy()).cuda()
            out, m = o.sort(t, n)
            o.check()
            torch.cuda.synchronize()
            got, exp = (out.cpu().numpy().view(dt)[:m] if m else np.empty(0, dt)), np.sort(a)
        elif name == "hostsort":   # clo_sort_with_host_data, around its pipelining threshold (128 MiB)
            n = int(rng.integers(1 << 20, (1 << 25) + 4097)) if et == "uint" else int(rng.integers(1 << 20, (1 << 24) + 4097))
            a = rng.integers(0, int(np.iinfo(dt).max), n, dtype=dt, endpoint=True)
            if rng.integers(0, 2):
                a >>= dt.type(int(rng.integers(1, 10)))
            got, exp = o.with_host_data(a, q), np.sort(a)
        elif name.startswith("scan"):
            a = rng.integers(0, 128, n).astype(dt)
            sdt = clo.api.CLO_TYPE_NP["ulong" if name == "scan64" else "uint"]
            src.write(q, a)
            o.with_device_data(q, src, dst, n)
            got = dst.read(q, sdt, n)
            exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(a.astype(np.uint64))[:-1])).astype(sdt)
        else:
            info = np.iinfo(dt)
            mode = int(rng.integers(0, 4))
            if mode == 0:
                a = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64 if info.min < 0 else np.uint64).astype(dt)
            elif mode == 1:    # few distinct values
                a = rng.integers(0, 7, n).astype(dt)
            elif mode == 2:    # sorted already
                a = np.sort(rng.integers(0, int(info.max) + 1, n, dtype=np.uint64).astype(dt))
            else:              # a narrow range
                a = (rng.integers(0, 1 << 10, n) + (int(info.max) >> 3)).astype(dt)
            src.write(q, a)
            in_place = bool(rng.integers(0, 2))
            o.with_device_data(q, src, None if in_place else dst, n)
            got = (src if in_place else dst).read(q, dt, n)
            exp = np.sort(a)
        calls += 1
        if not np.array_equal(got, exp):
            bad += 1
            print("MISMATCH", name, et, n, flush=True)
        if time.time() - last_print > 30:
            last_print = time.time()
            print("... %d calls, %d bad" % (calls, bad), flush=True)
    print("soak: %d calls in %.0f s, %d mismatches" % (calls, time.time() - t0, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
