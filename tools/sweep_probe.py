"""Single-sweep radix passes (clo_hip_radix1.hip) against the chain-free pair passes, and the
s_memtime breakdown of a sweep tile. GPU box only.
usage: python tools/sweep_probe.py [log2n] [u32|u64]"""
import os
import sys
import ctypes as C

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd import _hip  # noqa: E402
from cl_ops_amd._hip import lib  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
kind = sys.argv[2] if len(sys.argv) > 2 else "u32"
n = 1 << log2n
es = 4 if kind == "u32" else 8
rng = np.random.default_rng(0)
host = rng.integers(0, 1 << 32, n, dtype=np.uint32) if es == 4 else \
    rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).cuda()
dst = torch.empty_like(src)
ctx = clo.Context(0)
q = clo.Queue(ctx)
bs, bd = clo.Buffer(ctx, n * es, device_ptr=src.data_ptr()), clo.Buffer(ctx, n * es, device_ptr=dst.data_ptr())
ref = np.sort(host)


def run(mode, steps=20):
    os.environ["CLO_RADIX_SWEEP"] = mode
    s = clo.Sorter("satradix", ctx, "uint" if es == 4 else "ulong")
    for _ in range(4):
        s.with_device_data(q, bs, bd, n)
    q.finish()
    ok = np.array_equal(dst.cpu().numpy().view(host.dtype), ref)
    t = clo.HipEventTimer(q)
    t.start()
    for _ in range(steps):
        s.with_device_data(q, bs, bd, n)
    t.stop()
    q.finish()
    ms = t.elapsed_ms() / steps
    lib.clo_hip_timing_reset()
    lib.clo_hip_timing_enable(1)
    for _ in range(5):
        s.with_device_data(q, bs, bd, n)
    q.finish()
    lib.clo_hip_timing_enable(0)
    parts = {}
    for lab in ("radix_ghist", "radix_sweep", "radix_hist", "radix_offsets", "radix_pass"):
        c, tot = _hip.timing_read(lab)
        if c:
            parts[lab] = round(tot / c, 4)
    print("sweep=%s chunk_log=%s  %s 2^%d: %.3f ms/sort  %.0f Mkeys/s  correct=%s  per launch (ms): %s"
          % (mode, os.environ.get("CLO_R1_CHUNK_LOG", "-"), kind, log2n, ms, n / ms / 1e3, ok, parts), flush=True)
    return s


run("0").close()
for cl in ("3", "4"):
    os.environ["CLO_R1_CHUNK_LOG"] = cl
    s = run("1")
    if cl != "4":
        s.close()

# ---- stamps of the last pass's tiles ----
lib.clo_hip_radix_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
tile = 512 * (16 if es == 4 else 8)
tiles = (n + tile - 1) // tile
st = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
lib.clo_hip_radix_debug_stamps(st.data_ptr(), tiles)
s.with_device_data(q, bs, bd, n)
q.finish()
lib.clo_hip_radix_debug_stamps(None, 0)
a = st.cpu().numpy().reshape(tiles, 8)
t = a[:, :7].astype(np.float64)
names = ["ticket + load issue", "load latency + split 1", "-", "split 2 (+ row adds, publish)", "look-back (requests, wait)", "scatter issue"]
d = np.diff(t, axis=1)
clk = 100e6   # s_memtime ticks at the shader clock? printed raw: convert with the kernel's duration below
life = t[:, 6] - t[:, 0]
print("tiles %d; stamps in s_memtime ticks (median / mean / p90 per phase):" % tiles)
for k, nm in enumerate(names):
    print("  %-30s %8.0f %8.0f %8.0f   %4.1f %% of a tile's life" % (nm, np.median(d[:, k]), d[:, k].mean(), np.percentile(d[:, k], 90),
                                                                   100 * d[:, k].mean() / life.mean()))
print("  tile life            %8.0f %8.0f %8.0f" % (np.median(life), life.mean(), np.percentile(life, 90)))
span = t[:, 6].max() - t[:, 0].min()
print("  kernel span %.0f ticks; tiles in flight on average %.0f" % (span, life.sum() / span))
xcc = (a[:, 7] >> 32) & 0xF
blk = a[:, 7] & 0xFFFFFFFF
print("  XCC of tile's work-group == (chunk %% 8): %.3f ; blockIdx %% 8 == XCC for %.3f of the work-groups"
      % (np.mean(xcc == ((np.arange(tiles) >> 4) & 7)), np.mean((blk & 7) == (blk[0] & 7) + 0 * blk) if False else
         np.mean(((blk - xcc) & 7) == ((blk[0] - xcc[0]) & 7))))
s.close()
