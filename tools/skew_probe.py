"""ms per sort of 2^28 keys for key distributions other than uniform (device-resident, back-to-back
sorts, both radix paths): is there a cliff? GPU box only.
usage: python tools/skew_probe.py [log2n] [u32|u64]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLO_NO_WARMUP", "1")
import cl_ops_amd as clo  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
kind = sys.argv[2] if len(sys.argv) > 2 else "u32"
n = 1 << log2n
bits = 32 if kind == "u32" else 64
es = bits // 8
ctx = clo.Context(0)
q = clo.Queue(ctx)
g = torch.Generator(device="cuda")
g.manual_seed(1)


def rnd(hi):
    return torch.randint(0, hi, (n,), generator=g, device="cuda", dtype=torch.int64)


def wide():   # uniform over all `bits` bits, as an int64 bit pattern
    return rnd(1 << 32) if bits == 32 else (rnd(1 << 32) << 32) | rnd(1 << 32)


REP = 0x11111111 if bits == 32 else 0x1111111111111111
CASES = {
    "uniform %d bits" % bits: wide,
    "all keys equal": lambda: torch.full((n,), 0x12345678, device="cuda", dtype=torch.int64),
    "8 distinct keys": lambda: rnd(8) * (REP & 0x7FFFFFFFFFFFFFFF if bits == 64 else REP),
    "already sorted": lambda: torch.arange(n, device="cuda", dtype=torch.int64) * ((1 << (bits - 1)) // n),
    "reverse sorted": lambda: (n - 1 - torch.arange(n, device="cuda", dtype=torch.int64)) * ((1 << (bits - 1)) // n),
    "low 8 bits only": lambda: rnd(256),
    "high 8 bits only": lambda: rnd(128) << (bits - 8),
    "one hot digit (90 % in one bin of every digit)": lambda: torch.where(rnd(10) > 0, torch.full((n,), 0x7777777777777777 >> (64 - bits) if bits == 64 else 0x77777777, device="cuda", dtype=torch.int64), wide()),
}
for name, make in CASES.items():
    src = make()
    if bits == 32:
        src = src & 0xFFFFFFFF
        data = (src - ((src >> 31) << 32)).to(torch.int32)   # same bit pattern as uint32
    else:
        data = src
    del src
    dst = torch.empty_like(data)
    bs, bd = clo.Buffer(ctx, n * es, device_ptr=data.data_ptr()), clo.Buffer(ctx, n * es, device_ptr=dst.data_ptr())
    torch.cuda.synchronize()
    out = []
    for mode in ("0", "1"):
        os.environ["CLO_RADIX_SWEEP"] = mode
        s = clo.Sorter("satradix", ctx, "uint" if bits == 32 else "ulong")
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
        if bits == 32:
            u = dst.to(torch.int64) & 0xFFFFFFFF
            ok = bool((u[1:] >= u[:-1]).all())
        else:   # unsigned order of int64 bit patterns: flip the sign bit
            u = dst ^ (-(1 << 63))
            ok = bool((u[1:] >= u[:-1]).all())
        del u
        if not ok:
            print("NOT SORTED:", name, mode)
            sys.exit(1)
    print("%-50s pair passes %8.3f ms   single-sweep passes %8.3f ms" % (name, out[0], out[1]), flush=True)
    bs.close()
    bd.close()
    del data, dst
