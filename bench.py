#!/usr/bin/env python3
"""bench.py — headline benchmark of the sort/scan hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mkeys/s sorting 2^28 uint32 keys with satradix
(radix = 16, the reference's default: 4-bit digits), key order bit-exact vs the
reference algorithm. One "step" = one full sort of a fresh (unsorted) resident
array: clo_sort_with_device_data(src -> dst) through the C-ABI of
libcl_ops_hip.so; inputs are in HBM before the timed region starts (the
reference times the exec queue only: clo_sort_bench.c:160-162,201-207).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak
scaling, 2^28 keys per GPU; a step is the distributed sort of the 2^28*N-key
array: MSD bucket histogram -> count all-gather -> local partition ->
all-to-all(v) over xGMI -> local satradix (cl_ops_amd/multigpu.py).
value = all keys of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line. Besides the contract's fields it carries
  roofline:     the dominant kernel (the pass kernel) against the HBM peak:
                ALGORITHMIC bytes per launch / average launch duration measured
                live with HIP events on the kernel's own stream. SURVEY.md §8d
                prices a digit at 3*s bytes per element (read for histogram +
                read for scatter + write); the histogram read is a kernel of its
                own here, so the pass kernel is credited with the other two
                streams, and a launch handles TWO 4-bit digits of every
                element: 2 digits * 2*s * N. "moved_GBps" is what the kernel
                must move at least (one read + one write per launch),
                "traffic" what the PMC counters saw, and
                "algorithmic_GBps_per_step" the whole sort at the contract's
                3*s per digit (all kernels).
  cpu_baseline: the CPU oracle (a port of the reference decomposition,
                oracle/clo_oracle.c, OpenMP) on a bounded sample of the same
                workload, on this box's host cores.
Other workloads (parity-test configs, not bench lines): --workload
{satradix_pairs,satradix_u64,scan,abitonic,sbitonic}.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM = 3
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="satradix_u32",
                    choices=["satradix_u32", "satradix_pairs", "satradix_u64", "scan", "abitonic", "sbitonic"])
    ap.add_argument("--log2n", type=int, default=None, help="log2 of elements per GPU (default: BASELINE size)")
    ap.add_argument("--radix", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log2n", type=int, default=None)
    return ap.parse_args()


WORKLOADS = {
    # name: (default log2 n, elem type, description)
    "satradix_u32": (28, "uint", "satradix sort of 2^28 uint32 keys, radix=16 (4-bit digits)"),
    "satradix_pairs": (28, "ulong", "satradix sort of 2^28 (uint32 key, uint32 value) pairs, radix=16"),
    "satradix_u64": (28, "ulong", "satradix sort of 2^28 uint64 keys, radix=16"),
    "scan": (26, "uint", "blelloch exclusive scan of 2^26 uint32 (uint32 sums)"),
    "abitonic": (26, "uint", "abitonic sort of 2^26 uint32 keys"),
    "sbitonic": (16, "uint", "sbitonic sort of 2^16 uint32 keys"),
}


def make_input(workload, n, seed):
    rng = np.random.default_rng(seed)
    if workload in ("satradix_u32", "abitonic", "sbitonic"):
        return rng.integers(0, 1 << 32, n, dtype=np.uint32)
    if workload == "satradix_pairs":  # key in the high word, value = original index
        keys = rng.integers(0, 1 << 32, n, dtype=np.uint64)
        return (keys << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    if workload == "satradix_u64":
        return rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
    if workload == "scan":  # clo_scan_bench.c:219-223: values in [0,128)
        return rng.integers(0, 128, n, dtype=np.uint32)
    raise ValueError(workload)


def algorithmic_bytes_per_elem(workload, radix):
    """(per-launch bytes of the dominant kernel, per-step bytes), per element."""
    bits = int(np.log2(radix))
    if workload == "satradix_u32":
        return 3 * 4, (32 // bits) * 3 * 4
    if workload == "satradix_pairs":
        return 3 * 8, (32 // bits) * 3 * 8
    if workload == "satradix_u64":
        return 3 * 8, (64 // bits) * 3 * 8
    if workload == "scan":
        return 8, 8
    if workload == "abitonic":
        return 8, 8 * 58     # yardstick: G_ref = 58 global round trips of the reference strategy at 2^26
    if workload == "sbitonic":
        return 8, 8 * 136
    raise ValueError(workload)


DOMINANT_KERNEL = {"satradix_u32": "radix_pass", "satradix_pairs": "radix_pass", "satradix_u64": "radix_pass",
                   "scan": "scan", "abitonic": "bitonic_tile", "sbitonic": "bitonic_step"}


def cpu_baseline(workload, host_input, radix, sample_log2n):
    """Times the CPU oracle (port of the reference decomposition) on a bounded
    sample of the same input, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = min(len(os.sched_getaffinity(0)), 16)  # the box's CPU share for one GPU
    m = min(host_input.size, 1 << sample_log2n)
    sample = host_input[:m]
    t0 = time.perf_counter()
    if workload == "satradix_u32":
        out = O.satradix(sample, radix=radix, dev_max_lws=256, threads=cores)
        ok = bool(np.all(out[:-1] <= out[1:]))
    elif workload == "satradix_pairs":
        out = O.satradix(sample, radix=radix, dev_max_lws=256, threads=cores, key_size=4, key_shift=32)
        ok = bool(np.all((out[:-1] >> np.uint64(32)) <= (out[1:] >> np.uint64(32))))
    elif workload == "satradix_u64":
        out = O.satradix(sample, radix=radix, dev_max_lws=256, threads=cores)
        ok = bool(np.all(out[:-1] <= out[1:]))
    elif workload == "scan":
        out = O.blelloch(sample, np.uint32, dev_max_lws=256, threads=cores)
        ok = bool(out[1] == sample[0])
    else:
        t0 = time.perf_counter()
        out, _ = O.abitonic(sample, dev_max_lws=256) if workload == "abitonic" else (O.sbitonic(sample), 0)
        cores = 1
        ok = bool(np.all(out[:-1] <= out[1:]))
    dt = time.perf_counter() - t0
    unit = "MValues/s" if workload == "scan" else "Mkeys/s"
    return {"value": round(m / dt / 1e6, 3), "unit": unit, "cores": cores, "kind": "port",
            "sample": "first 2^%d elements of the same input, oracle/clo_oracle.c %s, %.1f s%s"
                      % (int(np.log2(m)), "OpenMP" if cores > 1 else "serial", dt, "" if ok else " (CHECK FAILED)")}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        args.gpus = world

    # The library loads its kernels at sorter creation with two small dummy sorts;
    # this script has untimed steps of its own for that, and the dummy launches
    # would dilute the per-kernel averages of a rocprofv3 run of this command.
    os.environ.setdefault("CLO_NO_WARMUP", "1")
    import torch
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    workload = args.workload
    log2n = args.log2n or WORKLOADS[workload][0]
    n = 1 << log2n
    etype = WORKLOADS[workload][1]
    es = 4 if etype == "uint" else 8
    if world > 1 and workload not in ("satradix_u32", "satradix_u64"):
        raise SystemExit("multi-GPU runs shard the satradix key sorts only; %s is replicas-only" % workload)

    host = make_input(workload, n, args.seed + rank)
    tdt = torch.int32 if es == 4 else torch.int64
    src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).to("cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()

    ctx = clo.Context(local_rank)
    # our own queue = our own HIP stream; every kernel of the path runs on it
    q = clo.Queue(ctx, profiling=False)
    bsrc = clo.Buffer(ctx, n * es, device_ptr=src.data_ptr())
    bdst = clo.Buffer(ctx, n * es, device_ptr=dst.data_ptr())

    sharded = None
    if workload.startswith("satradix"):
        kw = {}
        if workload == "satradix_pairs":
            kw = dict(key_type="uint", get_key="(uint) ((x) >> 32)")
        op = clo.Sorter("satradix", ctx, etype, options="radix=%d" % args.radix, **kw)
        if world > 1:
            from cl_ops_amd.multigpu import HipLocalOps, ShardedSorter
            sharded = ShardedSorter(HipLocalOps(etype, local_rank))
    elif workload == "scan":
        op = clo.Scanner("blelloch", ctx, "uint", "uint")
    else:
        op = clo.Sorter(workload, ctx, "uint")

    def step():
        if sharded is not None:
            return sharded.sort(src, n)          # the shard is only read: histogram + partition into the send buffer
        if workload == "scan":
            op.with_device_data(q, bsrc, bdst, n)
        else:
            op.with_device_data(q, bsrc, bdst, n)  # src stays unsorted, result in dst
        return None

    def fence():
        q.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # PREWARM untimed steps first (allocations of the cached aux buffers, clock
    # ramp), then the W warm-up steps the caller asked for
    for _ in range(PREWARM + args.warmup):
        step()
    fence()

    # ---- timed region: exactly K steps, fenced on both sides ----
    timer = clo.HipEventTimer(q)
    t0 = time.perf_counter()
    timer.start()
    last = None
    for _ in range(args.steps):
        last = step()
    timer.stop()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = timer.elapsed_ms() if sharded is None else None

    # ---- roofline leg: the same K steps again with every kernel launch
    # bracketed by HIP events on its own stream (clo_hip_timing_*). Kept out of
    # the timed region because an event pair between back-to-back kernels costs
    # a few microseconds of pipeline bubble (~4 % of a 5 ms sort). ----
    lib.clo_hip_timing_reset()
    lib.clo_hip_timing_enable(1)
    for _ in range(args.steps):
        step()
    fence()
    lib.clo_hip_timing_enable(0)

    t_max = wall
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())

    # ---- correctness of what was timed (size-independent properties) ----
    ok = True
    if sharded is not None:
        out_t, m = last
        udt = np.uint32 if es == 4 else np.uint64
        got = out_t[:m].cpu().numpy().view(udt)
        ok = bool(np.all(got[:-1] <= got[1:])) if m > 1 else True
        # per rank: [first key, last key, count, xor of output keys, xor of input keys] as raw 64-bit words
        mine = np.array([int(got[0]) if m else 0, int(got[-1]) if m else 0, m,
                         int(np.bitwise_xor.reduce(got)) if m else 0, int(np.bitwise_xor.reduce(host))],
                        dtype=np.uint64)
        t_mine = torch.from_numpy(mine.view(np.int64).copy()).cuda()
        alls = [torch.empty_like(t_mine) for _ in range(world)]
        dist.all_gather(alls, t_mine)
        if rank == 0:
            a = torch.stack(alls).cpu().numpy().view(np.uint64)
            ok = ok and int(a[:, 2].sum()) == n * world                    # nothing lost
            ok = ok and all(a[i, 1] <= a[i + 1, 0] for i in range(world - 1) if a[i, 2] and a[i + 1, 2])  # rank order = key order
            ok = ok and int(np.bitwise_xor.reduce(a[:, 3])) == int(np.bitwise_xor.reduce(a[:, 4]))       # same multiset
    else:
        got = dst.cpu().numpy().view(host.dtype)
        if workload == "scan":
            ok = bool(np.array_equal(got, (np.cumsum(host, dtype=np.uint64) - host).astype(np.uint32)))
        elif workload == "satradix_pairs":
            k = got >> np.uint64(32)
            v = got & np.uint64(0xFFFFFFFF)
            ok = bool(np.all(k[:-1] <= k[1:])) and bool(np.all((k[:-1] != k[1:]) | (v[:-1] < v[1:])))  # stable
            ok = ok and int(np.bitwise_xor.reduce(got)) == int(np.bitwise_xor.reduce(host))
        else:
            ok = bool(np.all(got[:-1] <= got[1:]))
            ok = ok and int(np.bitwise_xor.reduce(got)) == int(np.bitwise_xor.reduce(host))
            ok = ok and int(got.sum(dtype=np.uint64)) == int(host.sum(dtype=np.uint64))

    if rank == 0:
        per_launch_B, per_step_B = algorithmic_bytes_per_elem(workload, args.radix)
        label = DOMINANT_KERNEL[workload]
        cnt, tot_ms = _hip.timing_read(label)
        avg_ms = tot_ms / cnt if cnt else float("nan")
        per_launch_elems = n  # every launch of the dominant kernel sweeps the local array once
        digits_per_launch = 1
        if workload.startswith("satradix") and cnt:
            # the pass kernel handles several digit steps per launch (two 4-bit digits at radix 16)
            key_bits = 64 if workload == "satradix_u64" else 32
            digits = key_bits // int(np.log2(args.radix))
            digits_per_launch = max(1, round(digits * args.steps / cnt))
            per_launch_B = 2 * es * digits_per_launch   # scatter read + write, per digit step
        achieved = per_launch_B * per_launch_elems / (avg_ms * 1e-3) if cnt else float("nan")
        moved = 2 * es * per_launch_elems / (avg_ms * 1e-3) if cnt and workload.startswith("satradix") else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        unit = "MValues/s" if workload == "scan" else "Mkeys/s"
        total_elems = n * world * args.steps
        out = {
            "metric": "Mkeys/s sorting 2^28 uint32 (satradix, 4-bit digits); achieved % of HBM roofline"
                      if workload == "satradix_u32" else WORKLOADS[workload][2],
            "value": round(total_elems / t_max / 1e6, 1),
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm": PREWARM,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if es == 4 else "u64",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[workload][2], "elements_per_gpu": n, "radix": args.radix,
                       "parallelism": "single GPU" if world == 1 else
                       "msd-bucket-exchange x%d (RCCL send/recv all-to-all) + local satradix" % world,
                       "api": "clo_sort_with_device_data" if workload != "scan" else "clo_scan_with_device_data"},
            "correct": ok,
            "device_ms_per_step": round(dev_ms / args.steps, 4) if dev_ms is not None else None,
            "algorithmic_GBps_per_step": round(per_step_B * n * world / (t_max / args.steps) / 1e9, 1),
            "roofline": {"bound": "hbm", "kernel": label, "launches": cnt,
                         "avg_launch_ms": round(avg_ms, 5),
                         "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_launch_B * per_launch_elems,
                         "digit_steps_per_launch": digits_per_launch,
                         "moved_GBps": round(moved / 1e9, 1) if moved else None,
                         "note": "per-launch durations from HIP events on the kernel's stream over %d extra steps "
                                 "run right after the timed region" % args.steps},
        }
        if not args.no_cpu_baseline and world == 1:
            # sized for about 10-30 core-seconds of CPU work
            default_sample = {"satradix_u32": 28, "satradix_pairs": 27, "satradix_u64": 26, "scan": 26,
                              "abitonic": 20, "sbitonic": 16}[workload]
            out["cpu_baseline"] = cpu_baseline(workload, host, args.radix,
                                               args.cpu_sample_log2n or default_sample)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
