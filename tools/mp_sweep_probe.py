"""Diagnostic: the single-sweep passes on the NULL stream, in one process and in two
processes sharing the GPU. GPU box only."""
import faulthandler
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")


def work(tag, stream0, n, et):
    faulthandler.enable()
    import torch
    import cl_ops_amd as clo
    torch.cuda.set_device(0)
    ctx = clo.Context(0)
    q = clo.Queue(ctx, stream=0) if stream0 else clo.Queue(ctx)
    dt = np.uint32 if et == "uint" else np.uint64
    a = np.random.default_rng(1).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
    t = torch.from_numpy(a.view(np.int32 if et == "uint" else np.int64).copy()).cuda()
    buf = clo.Buffer(ctx, n * a.itemsize, device_ptr=t.data_ptr())
    s = clo.Sorter("satradix", ctx, et)
    for k in range(3):
        s.with_device_data(q, buf, None, n)
        q.finish()
    ok = np.array_equal(t.cpu().numpy().view(dt), np.sort(a))
    print(tag, "stream0" if stream0 else "own stream", et, n, "ok" if ok else "WRONG", flush=True)
    s.close()


def mp_worker(rank, stream0, n, et):
    work("proc%d" % rank, stream0, n, et)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    for et in ("uint", "ulong"):
        for stream0 in (False, True):
            mp.spawn(mp_worker, args=(stream0, 300000, et), nprocs=1, join=True)
            print("-- one process done", flush=True)
            mp.spawn(mp_worker, args=(stream0, 300000, et), nprocs=2, join=True)
            print("-- two processes done", flush=True)
