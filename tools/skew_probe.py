"""ms per sort of 2^28 uint32 keys for key distributions other than uniform (device-resident,
back-to-back sorts, both radix paths): is there a cliff? GPU box only.
usage: python tools/skew_probe.py [log2n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
import cl_ops_amd as clo  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
ctx = clo.Context(0)
q = clo.Queue(ctx)
g = torch.Generator(device="cuda")
g.manual_seed(1)


def rnd(hi):
    return torch.randint(0, hi, (n,), generator=g, device="cuda", dtype=torch.int64)


CASES = {
    "uniform 32 bits": lambda: rnd(1 << 32),
    "all keys equal": lambda: torch.full((n,), 0x12345678, device="cuda", dtype=torch.int64),
    "8 distinct keys": lambda: rnd(8) * 0x11111111,
    "already sorted": lambda: torch.arange(n, device="cuda", dtype=torch.int64) * ((1 << 32) // n),
    "reverse sorted": lambda: (n - 1 - torch.arange(n, device="cuda", dtype=torch.int64)) * ((1 << 32) // n),
    "low 8 bits only": lambda: rnd(256),
    "high 8 bits only": lambda: rnd(256) << 24,
    "one hot digit (90 % in one bin of every digit)": lambda: torch.where(rnd(10) > 0, torch.full((n,), 0x77777777, device="cuda", dtype=torch.int64), rnd(1 << 32)),
}
for name, make in CASES.items():
    src = make().to(torch.int32 if False else torch.int64)
    src = (src & 0xFFFFFFFF).to(torch.int64)
    src32 = src.to(torch.int32) if False else (src - ((src >> 31) << 32)).to(torch.int32)   # same bit pattern as uint32
    del src
    dst = torch.empty_like(src32)
    bs, bd = clo.Buffer(ctx, n * 4, device_ptr=src32.data_ptr()), clo.Buffer(ctx, n * 4, device_ptr=dst.data_ptr())
    torch.cuda.synchronize()
    out = []
    for mode in ("0", "1"):
        os.environ["CLO_RADIX_SWEEP"] = mode
        s = clo.Sorter("satradix", ctx, "uint")
        for _ in range(2):
            s.with_device_data(q, bs, bd, n)
        q.finish()
        t = clo.HipEventTimer(q)
        t.start()
        for _ in range(5):
            s.with_device_data(q, bs, bd, n)
        t.stop()
        q.finish()
        out.append(t.elapsed_ms() / 5)
        s.close()
        u = dst.to(torch.int64) & 0xFFFFFFFF
        ok = bool((u[1:] >= u[:-1]).all())
        del u
        if not ok:
            print("NOT SORTED:", name, mode)
            sys.exit(1)
    print("%-50s pair passes %8.3f ms   single-sweep passes %8.3f ms" % (name, out[0], out[1]), flush=True)
    bs.close()
    bd.close()
    del src32, dst
