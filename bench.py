#!/usr/bin/env python3
"""bench.py — headline benchmark of the sort/scan hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mkeys/s sorting 2^28 uint32 keys with satradix
(radix = 16, the reference's default: 4-bit digits), key order bit-exact vs the
reference algorithm. One "step" = one full sort of a fresh (unsorted) resident
array: clo_sort_with_device_data(src -> dst) through the C-ABI of
libcl_ops_hip.so; inputs are in HBM before the timed region starts (the
reference times the exec queue only: clo_sort_bench.c:160-162,201-207).

N > 1: one rank per GPU over RCCL. Started either by the driver
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) or
by this script itself: with WORLD_SIZE unset, `python bench.py --gpus N` spawns
that very command as a child process BEFORE anything here touches a GPU and
exits with its code. A step is the distributed sort of the whole array behind the
C API (include/clo_shard.h, clo_shard_sort_with_device_data; --exchange torch: the
same steps driven from Python): MSD partition into ranks x slices sub-buckets ->
count all-gather (with every rank's status: ranks fail together) -> one
all-to-all(v) of ncclSend/ncclRecv per slice over xGMI, slice j + 1 travelling while
slice j is sorted -> local satradix. Three legs, K timed steps each
(--scaling both, the default):
  weak    2^28 uint32 keys per GPU       -> the line's `value` (scaling: weak)
  strong  2^28 uint32 keys in total      -> "strong": {...}  (BASELINE.json's
          metric read literally: the same array at 1/2/4/8 GPUs)
  u64     2^28 uint64 keys per GPU       -> "config5_u64": {...} (BASELINE
          config 5 is this leg at N = 8: 2^31 keys)
value = all keys of all ranks / max-over-ranks time. "ranks_seen" proves how
many ranks took part (world size and an all-reduce of ones); "phases_ms" splits
a step into partition / count exchange / key exchange (until the first slice is
there) / local sort (and the rest of the exchange beside it). Every leg also carries
"exchange": bytes a rank sends over xGMI per step, the device time from the first
all-to-all(v) to the end of the last, the rate, and that rate as a fraction of
(N - 1) links x 153 GB/s; and "local_sort_roofline": the bytes the local sorts move
through HBM (PMC-calibrated per key where profiles/traffic_*.json has them, else the
minimum the kernels must move) over the local-sort phase, as a fraction of 8 TB/s.
A rank that spends more than --watchdog seconds in one leg ends itself with exit code
3 (a fresh exit, never a re-exec), which makes the launcher end the other ranks.

Rank 0 prints ONE JSON line. Besides the contract's fields it carries
  roofline:     PHYSICAL. `kernels` lists every kernel family of a step with
                launches per step, mean launch duration (HIP events on the
                kernel's own stream, measured live over K extra steps right
                after the timed region) and the bytes a launch moves through
                HBM: the PMC-measured traffic where profiles/traffic_*.json has
                it (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction), never less
                than what the kernel must move at least. `frac` = those bytes /
                duration / 8 TB/s for the dominant kernel — a fraction of the
                chip's peak, <= 1. SURVEY.md §8d's accounting (3*s bytes per
                element and 4-bit digit, whatever the kernels really move) is
                reported next to it as `contract_A_*`; it exceeds 1 when a pass
                handles two digits per trip through HBM.
  cpu_baseline: the CPU oracle (a port of the reference decomposition,
                oracle/clo_oracle.c, OpenMP) on a bounded sample of the same
                workload, on this box's host cores.
Other workloads (parity-test configs, not bench lines): --workload
{satradix_pairs,satradix_u64,scan,abitonic,sbitonic}.

--dry-run (tests only): no GPU — gloo, CPU tensors and the numpy stand-in of
the device steps from tests/numpy_ops.py, tiny sizes; exercises launching,
rendezvous, the exchange plan and the JSON line. Its `value` is not a result.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM = 3
SHARD_SETTLE = 10   # untimed steps of a sharded leg before its warm-up (buffers, clocks, the adaptive slice count)
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s measured copy ceiling)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="satradix_u32",
                    choices=["satradix_u32", "satradix_pairs", "satradix_u64", "scan", "abitonic", "sbitonic"])
    ap.add_argument("--log2n", type=int, default=None, help="log2 of elements per GPU (default: BASELINE size)")
    ap.add_argument("--radix", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scaling", default="both", choices=["weak", "strong", "both"],
                    help="N > 1: which legs to run (both = weak + strong + the uint64 leg of config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="N = 1: the headline line only, without the short legs of the other BASELINE configs (profiler runs)")
    ap.add_argument("--cpu-sample-log2n", type=int, default=None)
    ap.add_argument("--exchange", default="c", choices=["torch", "c"],
                    help="N > 1: the sort through the library's own C API over RCCL (include/clo_shard.h: "
                         "clo_shard_sort_with_device_data -> ncclSend/ncclRecv; the default) or driven from Python "
                         "through torch.distributed (batched isend/irecv on RCCL)")
    ap.add_argument("--slices", type=int, default=0, choices=[0, 1, 2, 4, 8],
                    help="N > 1, --exchange c: sub-buckets per rank that travel while earlier ones are sorted (0 = the library's default)")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="N > 1: seconds a rank may spend in one leg before it ends itself with exit code 3 (a stuck collective)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / exchange-plan test on CPU (gloo); not a measurement")
    return ap.parse_args(argv)


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


def reference_bitonic_round_trips(log2n):
    """Global-memory round trips of the reference's abitonic strategy at work-group
    size 1024 (clo_sort_abitonic.c:58-313: step 1 -> abit_any; steps above
    sfs = min(12, log2(lws) + maxps) = 12 go four at a time (priv_4s16v); a stage at or
    below sfs finishes in one local/hybrid kernel). 58 at 2^26 = SURVEY.md §8d's G_ref."""
    sfs = 12
    return sum(1 if s <= sfs else 1 + -(-(s - sfs) // 4) for s in range(1, log2n + 1))


def contract_bytes_per_elem(workload, radix, log2n):
    """SURVEY.md §8d's algorithmic bytes of one step, per element (the contract's
    accounting, independent of what the kernels really move)."""
    bits = int(np.log2(radix))
    if workload == "satradix_u32":
        return (32 // bits) * 3 * 4
    if workload == "satradix_pairs":
        return (32 // bits) * 3 * 8
    if workload == "satradix_u64":
        return (64 // bits) * 3 * 8
    if workload == "scan":
        return 8
    if workload == "abitonic":
        return 8 * reference_bitonic_round_trips(log2n)
    if workload == "sbitonic":
        return 8 * (log2n * (log2n + 1) // 2)
    raise ValueError(workload)


DOMINANT_KERNEL = {"satradix_u32": "radix_pass", "satradix_pairs": "radix_pass", "satradix_u64": "radix_pass",
                   "scan": "scan", "abitonic": "bitonic_tile", "sbitonic": "bitonic_presort"}

# kernel families a step may launch (the labels of clo_hip_timing_read, include/clo_hip.h)
FAMILIES = {
    "satradix": ["radix_ghist", "radix_sweep", "radix_hist", "radix_offsets", "radix_pass", "radix_small"],
    "scan": ["scan"],
    "abitonic": ["bitonic_presort", "bitonic_tile", "bitonic_strided", "bitonic_strided2"],
    "sbitonic": ["bitonic_presort", "bitonic_tile", "bitonic_strided", "bitonic_strided2", "bitonic_step"],   # (the tiled schedule; CLO_SBITONIC_STEPS=1: bitonic_step)
}


def min_moved_bytes(label, n, es, radix, key_bits=None):
    """What one launch of the family must move through HBM at the very least
    (element streams + its counters), averaged over the launches of a sort, from the
    kernels' definitions."""
    bits = int(np.log2(radix))
    pass_bits = 2 * bits if bits <= 4 else bits
    key_bits = key_bits or 8 * es
    passes = -(-key_bits // pass_bits)
    big = es >= 4 and n * es >= ((32 if es == 8 else 256) << 20)   # clo_radix_big_tiles (clo_hip_radix_rank.h)
    tile = (1024 if big else 512) * (8 if es == 8 else 16)
    counters = -(-n // tile) * (1 << pass_bits) * 4          # one row of counters per tile
    sweep_counters = -(-n // (512 * (8 if es == 8 else 16))) * (1 << pass_bits) * 4   # (the sweeps keep 512-thread tiles)
    # big tiles: every pass but the last also writes one digit byte per element, and
    # every histogram but the first reads those bytes instead of the elements
    dig = big and passes > 1 and bits in (4, 8)               # clo_radix_digit_stream; radix 16 / 256 only
    hist_in = (n * es + (passes - 1) * n) / passes if dig else n * es
    pass_extra = (passes - 1) * n / passes if dig else 0
    return {
        "radix_hist": hist_in + counters,                     # read every element (or its digit byte), write the tile histograms
        "radix_offsets": 2 * counters,                        # histograms in, offsets out (chunk sums are noise)
        "radix_pass": 2 * n * es + 2 * counters + pass_extra,  # read + write every element, read both counter rows
        "radix_ghist": n * es,                                # one read of the source
        "radix_sweep": 2 * n * es + 2 * sweep_counters,       # read + write every element; publish + look back
        "radix_small": 2 * n * es,
        "scan": 2 * n * es,                                   # (uint32 -> uint32 workload)
        "bitonic_presort": 2 * n * es, "bitonic_tile": 2 * n * es, "bitonic_strided": 2 * n * es,
        "bitonic_strided2": 2 * n * es, "bitonic_step": 2 * n * es,
    }[label]


def load_traffic(workload):
    """PMC-measured HBM bytes per launch by kernel family, from the tracked summary of
    tools/collect_profiles.sh (separate --pmc FETCH_SIZE / WRITE_SIZE passes)."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    try:
        d = json.load(open(path))
    except Exception:
        return {}, None
    fams = dict(d.get("families") or {})
    if not fams and d.get("hbm_bytes_per_launch"):           # round-1 file: the dominant kernel only
        fams = {DOMINANT_KERNEL[workload]: {"hbm_bytes_per_launch": d["hbm_bytes_per_launch"]}}
    return fams, {"log2n": d.get("log2n"), "source": d.get("source"), "kernels_sha16": d.get("kernels_sha16"),
                  "source_head": d.get("source_head")}


def host_cores():
    """(threads to use, cores of the affinity mask, cgroup CPU quota or None). A GPU box hands a job a share of its cores
    through the cgroup's CPU quota while the affinity mask still shows every core of the machine (256 on the MI355X boxes,
    16 of them ours): threads beyond the quota only get throttled — 256 OpenMP threads on a 16-core quota ran the radix
    port at 36 Mkeys/s where 16 threads reach 165. All the cores we may use = min(affinity, quota)."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    cores = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return cores, affinity, quota


def cpu_baseline(workload, host_input, radix, sample_log2n):
    """Times the CPU oracle (port of the reference decomposition) on a bounded
    sample of the same input, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores, affinity, quota = host_cores()   # every core this process may USE ("all host cores, core count printed")
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
        out, _ = O.abitonic(sample, dev_max_lws=256, threads=cores) if workload == "abitonic" else (O.sbitonic(sample, threads=cores), 0)
        ok = bool(np.all(out[:-1] <= out[1:]))
    dt = time.perf_counter() - t0
    unit = "MValues/s" if workload == "scan" else "Mkeys/s"
    return {"value": round(m / dt / 1e6, 3), "unit": unit, "cores": cores, "kind": "port",
            "affinity_cores": affinity, "cgroup_cpu_quota": quota,
            "sample": "first 2^%d elements of the same input, oracle/clo_oracle.c %s, %.1f s%s"
                      % (int(np.log2(m)), ("OpenMP on all %d usable cores (affinity mask %d, cgroup quota %s)" % (cores, affinity, "none" if quota is None else "%.1f" % quota))
                         if cores > 1 else "serial", dt, "" if ok else " (CHECK FAILED)")}


# ----------------------------------------------------------------------------
# launching N ranks from a plain `python bench.py --gpus N`
# ----------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """No launcher started us: become the launcher. Nothing in this process has
    touched a GPU yet (no torch import, no library load), and the ranks are fresh
    child processes — never an exec of a process that initialised HIP."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # The ranks watch themselves (--watchdog per leg); this is the outer net: the launcher and its ranks run in a
    # process group of their own and are ended as a group when the whole job overruns every leg's limit.
    limit = 3 * args.watchdog + 600 if args.watchdog and args.watchdog > 0 else None
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        sys.stderr.write("bench.py: the %d-rank job did not end within %.0f s: ending its process group\n" % (args.gpus, limit))
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.wait()
        return 3


# ----------------------------------------------------------------------------
# the distributed legs (N > 1)
# ----------------------------------------------------------------------------

class _Backend:
    """What differs between the real run (RCCL, HIP) and --dry-run (gloo, numpy)."""

    def __init__(self, dry, local_rank):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.dry = torch, dist, dry
        self.device = "cpu" if dry else torch.device("cuda", local_rank)
        if not dry:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=self.device)

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        self.dist.barrier()
        self.sync()

    def ops(self, etype, local_rank):
        if self.dry:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from numpy_ops import NumpyLocalOps          # test infrastructure, --dry-run only
            return NumpyLocalOps(etype)
        from cl_ops_amd.multigpu import HipLocalOps
        return HipLocalOps(etype, local_rank)

    def max_over_ranks(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


XGMI_LINK_GBPS = 153.0   # per link and direction pair (7 links per GPU; the task's figure)


def local_sort_bytes(m, es, radix):
    """HBM bytes one local satradix sort of m keys moves at the least (histogram + counter scan +
    pass kernels of every pass: min_moved_bytes per launch x launches)."""
    if m <= 0:
        return 0
    bits = int(np.log2(radix))
    passes = -(-8 * es // (2 * bits if bits <= 4 else bits))
    return int(sum(min_moved_bytes(k, m, es, radix) for k in ("radix_hist", "radix_offsets", "radix_pass")) * passes)


class _Watchdog:
    """Ends THIS rank (os._exit(3): no cleanup, no re-exec) when a leg takes longer than `seconds` — a
    collective some rank never joined would otherwise hold the whole job until the caller's own limit;
    the launcher ends the other ranks when one of them has gone."""

    def __init__(self, seconds, what):
        import threading
        self.t = threading.Timer(seconds, self._fire, args=(seconds, what)) if seconds and seconds > 0 else None
        if self.t:
            self.t.daemon = True
            self.t.start()

    @staticmethod
    def _fire(seconds, what):
        sys.stderr.write("bench.py: watchdog: %s still running after %.0f s — a stuck collective? ending this rank (exit 3)\n" % (what, seconds))
        sys.stderr.flush()
        os._exit(3)

    def done(self):
        if self.t:
            self.t.cancel()


def run_sharded_leg(be, etype, n_local, steps, warmup, seed, rank, world, local_rank, radix, exchange="c", slices=0, watchdog=0.0,
                    name="leg"):
    """K timed distributed sorts of n_local keys per rank; returns the leg's record
    (on every rank; only rank 0's is printed)."""
    wd = _Watchdog(watchdog, "rank %d, leg %s" % (rank, name))
    try:
        return _run_sharded_leg(be, etype, n_local, steps, warmup, seed, rank, world, local_rank, radix, exchange, slices, watchdog, name)
    finally:
        wd.done()


def _run_sharded_leg(be, etype, n_local, steps, warmup, seed, rank, world, local_rank, radix, exchange, slices, watchdog, name):
    torch, dist = be.torch, be.dist
    from cl_ops_amd.multigpu import ShardedSorter
    es = 4 if etype == "uint" else 8
    workload = "satradix_u32" if es == 4 else "satradix_u64"
    host = make_input(workload, n_local, seed + rank)
    src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).to(be.device)
    sharded, fallback = None, None
    if exchange == "c" and not be.dry:
        # The C driver over RCCL (the product path). It has never run on more than one RCCL rank before the first
        # multi-GPU run, so it has to prove itself before the legs depend on it: construction, then ONE small probe sort
        # (2^22 keys per rank, the smallest size that travels in slices: count all-gather + sliced send/recv on the second stream + slice sorts)
        # under a short watchdog of its own, then an all-reduce(MIN) of "mine worked". Any rank that failed takes every
        # rank to the Python driver TOGETHER, and the line says so: a scaling curve on the second path beats no curve.
        # A probe that hangs inside RCCL cannot be agreed upon: its watchdog ends the rank (exit 3, a plain exit).
        from cl_ops_amd.multigpu import CShardedSorter
        probe_wd = _Watchdog(min(watchdog, 120.0) if watchdog and watchdog > 0 else 0.0, "rank %d, C-driver probe sort of leg %s" % (rank, name))
        try:
            sharded = CShardedSorter(etype, local_rank, options="radix=%d%s" % (radix, ",slices=%d" % slices if slices else ""))
            pn = 1 << 22
            ph = make_input(workload, pn, seed + 977 + rank)
            pt = torch.from_numpy(ph.view(np.int32 if es == 4 else np.int64)).to(be.device)
            probe = CShardedSorter(etype, local_rank, options="radix=%d,slices=2,slice_min=1" % radix, transport=sharded.transport)
            try:
                po, pm = probe.sort(pt, pn)
                probe.check()
                be.sync()
                pg = po[:pm].cpu().numpy().view(ph.dtype)
                if pm > 1 and not bool(np.all(pg[:-1] <= pg[1:])):
                    raise RuntimeError("the probe sort's bucket is not sorted")
            finally:
                probe.close()
        except Exception as e:      # noqa: BLE001
            fallback = "%s: %s" % (type(e).__name__, e)
            if sharded is not None:
                try:
                    sharded.close()
                except Exception:   # noqa: BLE001
                    pass
                sharded = None
        finally:
            probe_wd.done()
        have = torch.tensor([0 if sharded is None else 1], dtype=torch.int64, device=be.device)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) == 0:
            if sharded is not None:
                sharded.close()
                sharded = None
            fallback = fallback or "another rank's C driver failed its probe sort"
            exchange = "torch (fallback from c: %s)" % fallback[:200]
    if sharded is None:
        sharded = ShardedSorter(be.ops(etype, local_rank))
        if be.dry:
            exchange = "torch over gloo (dry-run)"

    def step():
        return sharded.sort(src, n_local)      # the shard is only read: partition into the send buffer

    # (the C driver's adaptive slice count tries 4, 2, 1 and 8 slices twice each before it settles, and learns a call's
    # time two calls later: SHARD_SETTLE untimed steps in front of the W the caller asked for)
    for _ in range(SHARD_SETTLE + warmup):
        step()
    be.fence()
    # ---- timed region: exactly K steps, fenced on both sides ----
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    be.fence()
    wall = time.perf_counter() - t0
    t_max = be.max_over_ranks(wall)
    if hasattr(sharded, "check"):
        sharded.check()          # (outside the timed region) did a local sort give up a look-back spin? raises if so
    elif hasattr(sharded.ops, "check"):
        sharded.ops.check()

    # ---- phases: K more steps with device events around each phase (outside the timed region) ----
    sharded.phase_times = {}
    for _ in range(steps):
        step()
    be.fence()
    phases = {k: be.max_over_ranks(v / steps) for k, v in sorted(sharded.collect_phase_times().items())}
    sharded.phase_times = None
    # ---- the key exchange of the last step: bytes over xGMI, device time, rate (max / min over ranks) ----
    if hasattr(sharded, "ss"):                       # the C driver measures its own exchange (all slices)
        x = sharded.ss.exchange()
        x_bytes, x_s, x_slices = x["bytes_out"], x["ms"] * 1e-3, x["slices"]
    else:                                            # the Python driver: one exchange = the key_exchange phase
        x_bytes, x_s, x_slices = getattr(sharded, "last_exchange_bytes", 0), phases.get("key_exchange", 0.0), 1
    x_s_max = be.max_over_ranks(x_s)
    x_bytes_max = int(be.max_over_ranks(float(x_bytes)))
    x_rate = x_bytes_max / x_s_max / 1e9 if x_s_max > 0 else 0.0

    # ---- correctness of what was timed (size-independent properties) ----
    out_t, m = last
    udt = np.uint32 if es == 4 else np.uint64
    got = out_t[:m].cpu().numpy().view(udt)
    ok = bool(np.all(got[:-1] <= got[1:])) if m > 1 else True
    # per rank: [first key, last key, count, xor of output keys, xor of input keys, sum out, sum in] as raw 64-bit words
    mine = np.array([int(got[0]) if m else 0, int(got[-1]) if m else 0, m,
                     int(np.bitwise_xor.reduce(got)) if m else 0, int(np.bitwise_xor.reduce(host)),
                     int(got.sum(dtype=np.uint64)) if m else 0, int(host.sum(dtype=np.uint64))], dtype=np.uint64)
    t_mine = torch.from_numpy(mine.view(np.int64).copy()).to(be.device)
    alls = [torch.empty_like(t_mine) for _ in range(world)]
    dist.all_gather(alls, t_mine)
    a = torch.stack(alls).cpu().numpy().view(np.uint64)
    ok = ok and int(a[:, 2].sum()) == n_local * world                              # nothing lost
    ok = ok and all(a[i, 1] <= a[i + 1, 0] for i in range(world - 1) if a[i, 2] and a[i + 1, 2])  # rank order = key order
    ok = ok and int(np.bitwise_xor.reduce(a[:, 3])) == int(np.bitwise_xor.reduce(a[:, 4]))         # same multiset (xor)
    ok = ok and int(a[:, 5].sum(dtype=np.uint64)) == int(a[:, 6].sum(dtype=np.uint64))             # same multiset (sum mod 2^64)
    ones = torch.ones(1, dtype=torch.int64, device=be.device)
    dist.all_reduce(ones)
    if hasattr(sharded, "close"):
        sharded.close()
    elif hasattr(sharded.ops, "close"):
        sharded.ops.close()
    del src
    ls_bytes = local_sort_bytes(int(a[:, 2].max()), es, radix)          # the fullest rank's bucket
    ls_s = phases.get("local_sort", 0.0)
    return {"value": round(n_local * world * steps / t_max / 1e6, 1), "unit": "Mkeys/s",
            "exchange_stats": {
                "bytes_out_per_gpu": x_bytes_max, "slices": x_slices, "device_ms": round(x_s_max * 1e3, 4),
                "key_exchange_GBps": float("%.6g" % x_rate),
                "xgmi_peak_GBps": round((world - 1) * XGMI_LINK_GBPS, 1),
                "xgmi_frac": float("%.6g" % (x_rate / ((world - 1) * XGMI_LINK_GBPS))) if world > 1 else None,
                "note": "bytes a rank sends to the other ranks per step (its own bucket stays on the GPU) / device time from the "
                        "first all-to-all(v) to the end of the last; peak = (N - 1) links x %.0f GB/s" % XGMI_LINK_GBPS},
            "local_sort_roofline": {
                "bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s", "bytes": ls_bytes,
                "achieved": round(ls_bytes / ls_s / 1e9, 3) if ls_s > 0 else None,
                "frac": round(ls_bytes / ls_s / HBM_PEAK, 7) if ls_s > 0 else None,
                "basis": "minimum bytes the local sort's kernels must move for the fullest rank's bucket / the local_sort "
                         "phase (with slices that phase also waits for later slices to arrive: a lower bound)"},
            "ms_per_step": round(t_max / steps * 1e3, 4), "elements_per_gpu": n_local, "elements_total": n_local * world,
            "dtype": "u32" if es == 4 else "u64", "radix": radix, "correct": ok, "exchange": exchange,
            "ranks_seen": {"world_size": dist.get_world_size(), "allreduce_of_ones": int(ones.item())},
            "largest_bucket_over_mean": round(float(a[:, 2].max()) / n_local, 4),
            "phases_ms": {k: round(v * 1e3, 4) for k, v in phases.items()}}


def main_sharded(args, world, rank, local_rank):
    if args.workload not in ("satradix_u32", "satradix_u64"):
        raise SystemExit("multi-GPU runs shard the satradix key sorts only; %s is replicas-only" % args.workload)
    os.environ.setdefault("CLO_NO_WARMUP", "1")
    be = _Backend(args.dry_run, local_rank)
    etype = WORKLOADS[args.workload][1]
    log2n = args.log2n or (12 if args.dry_run else WORKLOADS[args.workload][0])
    n = 1 << log2n
    wbits = world.bit_length() - 1
    legs = {}
    common = (args.steps, args.warmup, args.seed, rank, world, local_rank, args.radix, args.exchange, args.slices, args.watchdog)
    if args.scaling in ("weak", "both"):
        legs["weak"] = run_sharded_leg(be, etype, n, *common, name="weak")
    if args.scaling in ("strong", "both"):
        legs["strong"] = run_sharded_leg(be, etype, max(n >> wbits, 1), *common, name="strong")
    if args.scaling == "both" and etype == "uint":       # BASELINE config 5's shape: uint64 keys, the same count per GPU
        legs["config5_u64"] = run_sharded_leg(be, "ulong", n, *common, name="config5_u64")
    head = legs.get("weak") or legs["strong"]
    ok = all(l["correct"] for l in legs.values())
    if rank == 0:
        out = {
            "metric": "Mkeys/s sorting 2^28 uint32 (satradix, 4-bit digits) at 1/2/4/8 GPUs",
            "value": head["value"], "unit": "Mkeys/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm": SHARD_SETTLE, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if "weak" in legs else "strong", "vs_baseline": None, "dtype": head["dtype"],
            "data": "synthetic" if not args.dry_run else "dry-run (no GPU: gloo + numpy stand-ins; NOT a measurement)",
            "config": {"workload": "satradix sort of %d x 2^%d %s keys sharded over %d GPUs, radix=%d"
                                   % (world, int(np.log2(head["elements_per_gpu"])), "uint32" if etype == "uint" else "uint64",
                                      world, args.radix),
                       "elements_per_gpu": head["elements_per_gpu"], "radix": args.radix,
                       "parallelism": "msd-bucket-exchange x%d (%s send/recv all-to-all, %d slice(s)) + local satradix"
                                      % (world, "gloo" if args.dry_run else "RCCL", head["exchange_stats"]["slices"]),
                       "api": "clo_shard_sort_with_device_data (C API over RCCL)" if head["exchange"] == "c" else
                              "cl_ops_amd.multigpu.ShardedSorter over clo_hip_msd_partition + clo_sort_with_device_data (%s)" % head["exchange"]},
            "correct": ok, "ranks_seen": head["ranks_seen"], "phases_ms": head["phases_ms"],
            "exchange_stats": head["exchange_stats"], "roofline": head["local_sort_roofline"],
        }
        for k, v in legs.items():
            if v is not head:
                out[k] = v
        print(json.dumps(out), flush=True)
    be.dist.barrier()
    be.dist.destroy_process_group()
    return 0 if ok else 1


# ----------------------------------------------------------------------------
# one GPU
# ----------------------------------------------------------------------------

# the kernel sources a workload's launches come from (cl_ops_amd/csrc/hip/)
KERNEL_SOURCES = {
    "satradix": ["clo_hip_radix.hip", "clo_hip_radix4.hip", "clo_hip_radixw.hip", "clo_hip_radix1.hip", "clo_hip_radix_rank.h", "clo_hip_internal.h"],
    "scan": ["clo_hip_scan.hip", "clo_hip_internal.h"],
    "abitonic": ["clo_hip_bitonic.hip", "clo_hip_bitonic_impl.h", "clo_hip_bitonic_e4.hip", "clo_hip_internal.h"],
    "sbitonic": ["clo_hip_bitonic.hip", "clo_hip_bitonic_impl.h", "clo_hip_bitonic_e4.hip", "clo_hip_internal.h"],
}


def kernel_sources_sha16(group):
    """Content hash of the kernel sources of a workload's family: the GPU box has no .git, so this is how a line says
    which kernels it ran and whether profiles/traffic_*.json was collected on the same ones."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[group]:
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "cl_ops_amd", "csrc", "hip", name), "rb").read())
    return h.hexdigest()[:16]


def single_leg(workload, log2n, steps, warmup, radix, seed, cpu_sample_log2n=None, with_cpu=True):
    """One workload on cuda:0 through the C API: PREWARM + warmup untimed steps, exactly `steps` timed ones between
    fences, then the same number again with every kernel bracketed by HIP events (roofline + live guard), a
    size-independent correctness check of what was timed, and the CPU oracle on a bounded sample. Returns the
    full line as a dict."""
    import torch
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib

    n = 1 << log2n
    etype = WORKLOADS[workload][1]
    es = 4 if etype == "uint" else 8

    host = make_input(workload, n, seed)
    src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).to("cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()

    ctx = clo.Context(0)
    # our own queue = our own HIP stream; every kernel of the path runs on it
    q = clo.Queue(ctx, profiling=False)
    bsrc = clo.Buffer(ctx, n * es, device_ptr=src.data_ptr())
    bdst = clo.Buffer(ctx, n * es, device_ptr=dst.data_ptr())

    if workload.startswith("satradix"):
        kw = {}
        if workload == "satradix_pairs":
            kw = dict(key_type="uint", get_key="(uint) ((x) >> 32)")
        op = clo.Sorter("satradix", ctx, etype, options="radix=%d" % radix, **kw)
    elif workload == "scan":
        op = clo.Scanner("blelloch", ctx, "uint", "uint")
    else:
        op = clo.Sorter(workload, ctx, "uint")

    def step():
        op.with_device_data(q, bsrc, bdst, n)  # src stays unsorted, result in dst

    def fence():
        q.finish()
        torch.cuda.synchronize()

    # PREWARM untimed steps first (allocations of the cached aux buffers, clock
    # ramp), then the W warm-up steps the caller asked for
    for _ in range(PREWARM + warmup):
        step()
    fence()

    # ---- timed region: exactly K steps, fenced on both sides ----
    timer = clo.HipEventTimer(q)
    t0 = time.perf_counter()
    timer.start()
    for _ in range(steps):
        step()
    timer.stop()
    fence()
    wall = time.perf_counter() - t0
    dev_ms = timer.elapsed_ms()

    # ---- roofline leg: the same K steps again with every kernel launch
    # bracketed by HIP events on its own stream (clo_hip_timing_*). Kept out of
    # the timed region because an event pair between back-to-back kernels costs
    # a few microseconds of pipeline bubble (~4 % of a 5 ms sort). ----
    lib.clo_hip_timing_reset()
    lib.clo_hip_timing_enable(1)
    for _ in range(steps):
        step()
    fence()
    lib.clo_hip_timing_enable(0)

    # ---- correctness of what was timed (size-independent properties) ----
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
    del got

    # ---- the line ----
    group = "satradix" if workload.startswith("satradix") else workload
    traffic, tmeta = load_traffic(workload)
    if tmeta and tmeta.get("log2n") not in (None, log2n):
        traffic, tmeta = {}, None                      # measured at another size: not comparable
    sha_now = kernel_sources_sha16(group)
    stale = bool(traffic) and (tmeta or {}).get("kernels_sha16") != sha_now
    # launches per step as the PMC collection saw them (for the live guard below): counted BEFORE the bytes are used —
    # a stale file whose launches no longer match describes other kernels, and its bytes are not used at all
    # (a run with CLO_RADIX_SWEEP=1 forces the other path: its launches are the ones the forced collection saw)
    lkey = "sweep_launches_per_step" if os.environ.get("CLO_RADIX_SWEEP") == "1" else "launches_per_step"
    expected = {f: v[lkey] for f, v in traffic.items() if v.get(lkey)}
    seen_counts = {}
    for label in FAMILIES[group]:
        cnt, _ = _hip.timing_read(label)
        if cnt:
            seen_counts[label] = round(cnt / steps, 3)
    counts_ok = None if not expected else bool(set(expected) == set(seen_counts) and all(abs(seen_counts[f] - expected[f]) < 1e-6 for f in expected))
    traffic_dropped = bool(stale and counts_ok is False)
    if traffic_dropped:
        traffic = {}
    kernels = []
    for label in FAMILIES[group]:
        cnt, tot_ms = _hip.timing_read(label)
        if not cnt:
            continue
        avg_ms = tot_ms / cnt
        floor_b = min_moved_bytes(label, n, es, radix, 32 if workload == "satradix_pairs" else None)
        pmc_b = (traffic.get(label) or {}).get("hbm_bytes_per_launch")
        moved = max(floor_b, pmc_b or 0)
        kernels.append({"name": label, "launches_per_step": round(cnt / steps, 3), "avg_launch_ms": round(avg_ms, 5),
                        "bytes_per_launch": int(moved), "traffic_pmc": pmc_b, "min_moved_bytes": int(floor_b),
                        "GBps": round(moved / (avg_ms * 1e-3) / 1e9, 1),
                        "frac": round(moved / (avg_ms * 1e-3) / HBM_PEAK, 4)})
    label = DOMINANT_KERNEL[workload]
    if not any(k["name"] == label for k in kernels) and kernels:
        label = max(kernels, key=lambda k: k["avg_launch_ms"] * k["launches_per_step"])["name"]
    dom = next((k for k in kernels if k["name"] == label), None)
    step_bytes = sum(k["bytes_per_launch"] * k["launches_per_step"] for k in kernels)
    kernel_ms = sum(k["avg_launch_ms"] * k["launches_per_step"] for k in kernels)
    contract_B = contract_bytes_per_elem(workload, radix, log2n) * n
    unit = "MValues/s" if workload == "scan" else "Mkeys/s"
    ms_step = wall / steps * 1e3
    roof = {"bound": "hbm", "kernel": label, "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
    if workload in ("abitonic", "sbitonic") and n * es <= (256 << 20):
        # every pass re-reads what the pass before wrote, and the whole array fits the 256 MiB last-level cache:
        # the rate below is cache-assisted (it can exceed the part's 6.3 TB/s copy ceiling), not an HBM fraction
        roof["bound"] = "llc+hbm"
        roof["bound_note"] = ("the %d MiB array stays in the 256 MiB Infinity Cache between passes: frac is a rate relative to "
                              "the 8 TB/s HBM peak, not a fraction of HBM traffic" % ((n * es) >> 20))
    if dom:
        assert dom["frac"] <= 1.0, "a physical fraction above 1: %r" % (dom,)
        roof.update({
            "achieved": dom["GBps"], "frac": dom["frac"], "traffic": dom["traffic_pmc"],
            "launches": int(round(dom["launches_per_step"] * steps)), "avg_launch_ms": dom["avg_launch_ms"],
            "bytes_per_launch": dom["bytes_per_launch"],
            "basis": ("stale PMC" if stale else "PMC traffic") + " (profiles/traffic_%s.json)" % workload if dom["traffic_pmc"] else
                     ("minimum moved bytes (profiles/traffic_%s.json is stale AND counts other launches per step: not used)" % workload if traffic_dropped
                      else "minimum moved bytes (no PMC summary for this size in profiles/)"),
        })
    # ---- live guard: what a stale PMC file cannot vouch for is checked against THIS run: the launches a step
    # makes per kernel family against the ones the PMC collection saw, and the kernels' summed durations against
    # the step (a kernel family that appeared, vanished or doubled, or time spent outside the listed kernels) ----
    seen = {k["name"]: k["launches_per_step"] for k in kernels}
    ratio = kernel_ms / ms_step if ms_step > 0 else 0.0
    tol = 0.05 if ms_step >= 0.5 else (0.1 if ms_step >= 0.1 else 0.5)   # (launch-bound steps: the event pairs themselves are a share of the step)
    guard = {"launches_per_step": seen, "expected_launches_per_step": expected or None, "launch_counts_ok": counts_ok,
             "kernel_ms_per_step": round(kernel_ms, 4), "kernel_ms_over_step_ms": round(ratio, 4), "tolerance": tol,
             "timing_ok": bool(abs(ratio - 1.0) <= tol),
             "kernels_sha16": sha_now, "traffic_kernels_sha16": (tmeta or {}).get("kernels_sha16"),
             "traffic_source_head": (tmeta or {}).get("source_head"), "traffic_stale": stale if (traffic or traffic_dropped) else None,
             "traffic_dropped": traffic_dropped}
    # a file that claims to describe THESE kernels (same source hash) and counts other launches fails the run; a stale
    # file is labelled, and dropped when even its launches differ
    guard["ok"] = bool(guard["timing_ok"] and (counts_ok is not False or stale))
    roof.update({
        "kernels": kernels,
        "kernel_ms_per_step": round(kernel_ms, 4),                # sum over the list: what is left of ms_per_step is launch gaps
        "traffic_step_bytes": int(step_bytes),
        "step_GBps": round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
        "step_frac": round(step_bytes / (ms_step * 1e-3) / HBM_PEAK, 4),   # whole step, physical
        "contract_A_bytes_per_step": int(contract_B),              # SURVEY.md §8d's accounting: a LABEL (it prices 4-bit trips; the sort makes 8-bit ones), not a roofline
        "contract_A_GBps": round(contract_B / (ms_step * 1e-3) / 1e9, 1),
        "contract_A_frac": round(contract_B / (ms_step * 1e-3) / HBM_PEAK, 4),
        "note": "per-launch durations from HIP events on the kernel's stream over %d extra steps run right after the "
                "timed region; frac = bytes really moved / duration / 8 TB/s" % steps,
    })
    out = {
        "metric": "Mkeys/s sorting 2^28 uint32 (satradix, 4-bit digits); achieved % of HBM roofline"
                  if workload == "satradix_u32" else WORKLOADS[workload][2],
        "value": round(n * steps / wall / 1e6, 1), "unit": unit, "n_gpus": 1, "steps": steps,
        "warmup": warmup, "prewarm": PREWARM, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32" if es == 4 else "u64", "data": "synthetic",
        "config": {"workload": WORKLOADS[workload][2] if log2n == WORKLOADS[workload][0] else
                   WORKLOADS[workload][2].replace("2^%d" % WORKLOADS[workload][0], "2^%d" % log2n),
                   "elements_per_gpu": n, "radix": radix, "parallelism": "single GPU",
                   "api": "clo_sort_with_device_data" if workload != "scan" else "clo_scan_with_device_data"},
        "correct": ok, "device_ms_per_step": round(dev_ms / steps, 4), "roofline": roof, "live_guard": guard,
    }
    for x in (op, bsrc, bdst, q, ctx):
        x.close()
    del src, dst
    torch.cuda.empty_cache()
    if with_cpu:
        # sized for about 10-30 core-seconds of CPU work
        default_sample = {"satradix_u32": 28, "satradix_pairs": 27, "satradix_u64": 26, "scan": 26,
                          "abitonic": 24, "sbitonic": 16}[workload]
        out["cpu_baseline"] = cpu_baseline(workload, host, radix, min(cpu_sample_log2n or default_sample, log2n))
    return out


def shard_world1_leg(etype, log2n, steps, warmup, radix, seed, slices=0):
    """BASELINE config 5's code path on the one GPU there is: clo_shard_sort_with_device_data (include/clo_shard.h) on
    a one-rank RCCL communicator with `loopback=1` — partition, count all-gather, the slices' grouped
    ncclSend/ncclRecv (this rank to itself) on the transfer stream, the local sorts beside them. What a node adds is
    xGMI in place of a device copy."""
    import torch
    from cl_ops_amd.multigpu import CShardedSorter
    es = 4 if etype == "uint" else 8
    n = 1 << log2n
    workload = "satradix_u32" if es == 4 else "satradix_u64"
    host = make_input(workload, n, seed)
    src = torch.from_numpy(host.view(np.int32 if es == 4 else np.int64)).to("cuda")
    s = CShardedSorter(etype, 0, options="radix=%d,loopback=1%s" % (radix, ",slices=%d" % slices if slices else ""))
    try:
        for _ in range(SHARD_SETTLE + warmup):
            s.sort(src)
        s.check()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out_t, m = s.sort(src)
        s.check()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        # (the result buffer belongs to the object and is only valid until the next call: copy it out NOW, before the
        # phase loop sorts again — with adaptive slices or a grow it may even be reallocated)
        got = out_t[:m].cpu().numpy().view(host.dtype)
        s.phase_times = {}
        for _ in range(steps):
            s.sort(src)
        torch.cuda.synchronize()
        phases = {k: v / steps for k, v in s.collect_phase_times().items()}
        x = s.ss.exchange()
        ok = m == n and bool(np.all(got[:-1] <= got[1:]))
        ok = ok and int(np.bitwise_xor.reduce(got)) == int(np.bitwise_xor.reduce(host))
        ok = ok and int(got.sum(dtype=np.uint64)) == int(host.sum(dtype=np.uint64))
        ok = ok and bool(np.array_equal(src.cpu().numpy().view(host.dtype), host))      # the shard is only read
    finally:
        s.close()
    del src
    torch.cuda.empty_cache()
    ms_step = wall / steps * 1e3
    ls_bytes = local_sort_bytes(n, es, radix)
    ls_s = phases.get("local_sort", 0.0)
    return {"workload": "satradix of 2^%d %s keys through clo_shard_sort_with_device_data (C API) on ONE rank over RCCL, loopback=1"
                        % (log2n, "uint32" if es == 4 else "uint64"),
            "value": round(n * steps / wall / 1e6, 1), "unit": "Mkeys/s", "ms_per_step": round(ms_step, 4), "correct": ok, "steps": steps,
            "slices": x["slices"], "exchange_device_ms": round(x["ms"], 4),
            "phases_ms": {k: round(v * 1e3, 4) for k, v in sorted(phases.items())},
            # bytes a plain local sort of the same keys must move at least / the local_sort phase; the whole step adds the exchange's copy
            # (with slices > 1 the local_sort phase includes the stream's waits for the later slices: a lower bound of the sort's rate)
            "local_sort_frac": round(ls_bytes / ls_s / HBM_PEAK, 4) if ls_s > 0 else None,
            "step_frac": round((ls_bytes + 2 * n * es) / (ms_step * 1e-3) / HBM_PEAK, 4)}


# the BASELINE.json configs beside the headline, in its order: (key, workload, steps, warmup)
CONFIG_LEGS = (("1_sbitonic_2p16", "sbitonic", 50, 5), ("2_scan_2p26", "scan", 30, 5), ("3_abitonic_2p26", "abitonic", 10, 2),
               ("4_satradix_pairs_2p28", "satradix_pairs", 8, 2), ("5_satradix_u64_2p28_one_shard", "satradix_u64", 5, 2))


def leg_summary(full):
    """A config leg as it travels in the headline's line: what the verdict asked for (value, ms_per_step, correct,
    dominant-kernel frac, step_frac, cpu_baseline) and not much more — the line stays a few KB."""
    r = full["roofline"]
    basis = r.get("basis") or ""
    out = {"workload": full["config"]["workload"], "value": full["value"], "unit": full["unit"], "ms_per_step": full["ms_per_step"],
           "correct": full["correct"], "steps": full["steps"], "kernel": r.get("kernel"), "frac": r.get("frac"), "bound": r.get("bound"),
           "basis": "stale PMC" if basis.startswith("stale") else ("PMC" if basis.startswith("PMC") else "min bytes"),
           "step_frac": r.get("step_frac"), "guard_ok": full["live_guard"]["ok"]}
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"]}
    return out


def main_single(args):
    # The library loads its kernels at sorter creation with two small dummy sorts;
    # this script has untimed steps of its own for that, and the dummy launches
    # would dilute the per-kernel averages of a rocprofv3 run of this command.
    os.environ.setdefault("CLO_NO_WARMUP", "1")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(0)
    workload = args.workload
    log2n = args.log2n or WORKLOADS[workload][0]
    out = single_leg(workload, log2n, args.steps, args.warmup, args.radix, args.seed, args.cpu_sample_log2n, not args.no_cpu_baseline)
    ok = out["correct"]
    if not out["live_guard"]["ok"]:
        sys.stderr.write("bench.py: live guard: %r\n" % (out["live_guard"],))
    # ---- every other BASELINE config, one short leg each, in this process (no re-exec, no child after HIP is up) ----
    if workload == "satradix_u32" and args.log2n is None and not args.no_configs:
        t0 = time.perf_counter()
        configs = {}
        for key, w, k, wu in CONFIG_LEGS:
            try:
                configs[key] = leg_summary(single_leg(w, WORKLOADS[w][0], k, wu, args.radix, args.seed, None, not args.no_cpu_baseline))
            except Exception as e:      # noqa: BLE001 (a leg that fails says so; the headline stands)
                configs[key] = {"correct": False, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        try:
            configs["5_sharded_c_path_world1_rccl_u64_2p28"] = shard_world1_leg("ulong", 28, 5, 2, args.radix, args.seed, args.slices)
        except Exception as e:          # noqa: BLE001
            configs["5_sharded_c_path_world1_rccl_u64_2p28"] = {"correct": False, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        out["configs"] = configs
        out["configs_wall_s"] = round(time.perf_counter() - t0, 1)
        ok = ok and all(c.get("correct") for c in configs.values())
    print(json.dumps(out), flush=True)
    if out["live_guard"]["launch_counts_ok"] is False and not out["live_guard"]["traffic_stale"]:
        return 1
    return 0 if ok else 1


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0:                     # started as a plain script
        if args.gpus > 1:
            return self_launch(args)
        return main_single(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return main_single(args)
    return main_sharded(args, world, rank, local_rank)


if __name__ == "__main__":
    sys.exit(main())
