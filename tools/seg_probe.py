"""Developer probe (GPU box): per-kernel times of the segmented sort (clo_hip_radix_sort_segmented) against the plain
sort of the same keys. usage: python tools/seg_probe.py [log2n=28] [uint|ulong] [nseg=256] [key_bits=elem bits - 8]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
import cl_ops_amd as clo  # noqa: E402
from cl_ops_amd import _hip  # noqa: E402
from cl_ops_amd._hip import lib  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
etype = sys.argv[2] if len(sys.argv) > 2 else "uint"
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 256
es = 4 if etype == "uint" else 8
kb = int(sys.argv[4]) if len(sys.argv) > 4 else 8 * es - 8
n = 1 << logn
dt = np.uint32 if es == 4 else np.uint64
a = np.random.default_rng(0).integers(0, np.iinfo(dt).max, n, dtype=dt, endpoint=True)
# segments = the key ranges of the top log2(nseg) bits, as an MSD partition leaves them
top = (a >> dt(8 * es - int(np.log2(nseg)))).astype(np.int64)
order = np.argsort(top, kind="stable")
a = a[order]
seg_counts = np.bincount(top, minlength=nseg)
tdt = np.int32 if es == 4 else np.int64
src = torch.from_numpy(a.view(tdt).copy()).cuda()
ta, tb = torch.empty_like(src), torch.empty_like(src)
need = lib.clo_hip_radix_seg_workspace_bytes(n, nseg, es, 4)
ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
sc = (C.c_size_t * nseg)(*[int(x) for x in seg_counts])
in_b = C.c_int(0)
stream = torch.cuda.current_stream().cuda_stream
FAM = ("radix_seg_tables", "radix_hist", "radix_offsets", "radix_pass")


def seg():
    ta.copy_(src)
    _hip.check(lib.clo_hip_radix_sort_segmented(ta.data_ptr(), ta.data_ptr(), tb.data_ptr(), n, sc, nseg, None, None, None, 0, es, 0, kb, 4,
                                                ws.data_ptr(), need, stream, C.byref(in_b)))


for _ in range(3):
    seg()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(10):
    ta.copy_(src)
    e0.record()
    _hip.check(lib.clo_hip_radix_sort_segmented(ta.data_ptr(), ta.data_ptr(), tb.data_ptr(), n, sc, nseg, None, None, None, 0, es, 0, kb, 4,
                                                ws.data_ptr(), need, stream, C.byref(in_b)))
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
got = (tb if in_b.value else ta).cpu().numpy().view(dt)
exp = np.empty_like(a)
at = 0
for c in seg_counts:                      # every segment on its own, stably, by the low kb bits
    part = a[at:at + c]
    exp[at:at + c] = part[np.argsort(part & dt((1 << kb) - 1), kind="stable")]
    at += c
print("2^%d %s, %d segments, %d key bits: segmented sort %.3f ms%s" % (logn, etype, nseg, kb, tot / 10, "" if np.array_equal(got, exp) else "  WRONG"))
lib.clo_hip_timing_reset()
lib.clo_hip_timing_enable(1)
for _ in range(5):
    seg()
torch.cuda.synchronize()
lib.clo_hip_timing_enable(0)
for f in FAM:
    c, ms = _hip.timing_read(f)
    if c:
        print("   %-17s %5.2f launches per sort, %.4f ms each, %.4f ms per sort" % (f, c / 5, ms / c, ms / 5))

# the plain sort of the same keys on the same key bits
ctx = clo.Context(0)
q = clo.Queue(ctx, stream=stream)
s = clo.Sorter("satradix", ctx, etype, get_key="((x) & 0x%x)" % ((1 << kb) - 1)) if kb < 8 * es else clo.Sorter("satradix", ctx, etype)
bsrc, bdst = clo.Buffer(ctx, n * es, device_ptr=src.data_ptr()), clo.Buffer(ctx, n * es, device_ptr=tb.data_ptr())
for _ in range(3):
    s.with_device_data(q, bsrc, bdst, n)
torch.cuda.synchronize()
tot = 0.0
for _ in range(10):
    e0.record()
    s.with_device_data(q, bsrc, bdst, n)
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print("plain satradix on the low %d bits of the whole array: %.3f ms" % (kb, tot / 10))
lib.clo_hip_timing_reset()
lib.clo_hip_timing_enable(1)
for _ in range(5):
    s.with_device_data(q, bsrc, bdst, n)
torch.cuda.synchronize()
lib.clo_hip_timing_enable(0)
for f in FAM:
    c, ms = _hip.timing_read(f)
    if c:
        print("   %-17s %5.2f launches per sort, %.4f ms each, %.4f ms per sort" % (f, c / 5, ms / c, ms / 5))
