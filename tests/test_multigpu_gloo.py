"""world_size-2 (and 4, 8) gloo tests of the multi-GPU host logic
(cl_ops_amd/multigpu.py: count exchange, send/recv plan, P2P batch order,
capacity handling) on CPU tensors. The device-side steps are injected as an
oracle/numpy-backed LocalOps — test infrastructure only; the product's
HipLocalOps needs a GPU and is covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from numpy_ops import NumpyLocalOps  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, elem_type, n, skew, out_dir):
    import torch
    import torch.distributed as dist
    from cl_ops_amd.multigpu import ShardedSorter
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        np_dt = np.uint32 if elem_type == "uint" else np.uint64
        t_dt = torch.int32 if elem_type == "uint" else torch.int64
        rng = np.random.default_rng(100 + rank)
        hi = np.iinfo(np_dt).max
        if skew == "empty":   # a rank without keys, and every key of the others in bucket 0: ranks that send / receive nothing
            n = 0 if rank == world - 1 else n
            a = rng.integers(0, hi >> 4, n, dtype=np_dt, endpoint=True)
        elif skew:   # most keys in one bucket: exercises the receive-capacity path
            a = rng.integers(0, hi // 8, n, dtype=np_dt, endpoint=True)
            a[: n // 10] = rng.integers(0, hi, n // 10, dtype=np_dt, endpoint=True)
        else:
            a = rng.integers(0, hi, n, dtype=np_dt, endpoint=True)
        local = torch.from_numpy(a.view(np.int32 if elem_type == "uint" else np.int64).copy())
        ss = ShardedSorter(NumpyLocalOps(elem_type))
        out, m = ss.sort(local, n)
        got = out.numpy()[:m].view(np_dt).copy()
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), a)
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), got)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,elem_type,n,skew", [(2, "uint", 5000, False), (2, "ulong", 3000, False),
                                                     (4, "uint", 2000, False), (2, "uint", 4000, True),
                                                     (8, "uint", 1500, False), (8, "ulong", 700, True),    # the driver's N = 8 plan
                                                     (2, "uint", 3000, "empty"), (4, "ulong", 900, "empty")])
def test_sharded_sort_over_gloo(tmp_path, world, elem_type, n, skew):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, elem_type, n, skew, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / ("in_%d.npy" % r)) for r in range(world)]
    outs = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)]
    # rank order == global order, and it is exactly the sorted multiset
    assert np.array_equal(np.concatenate(outs), np.sort(np.concatenate(ins)))
    bits = 32 if elem_type == "uint" else 64
    b = world.bit_length() - 1
    for r, o in enumerate(outs):           # rank r holds exactly bucket r
        if o.size:
            assert np.all((o >> o.dtype.type(bits - b)) == r)


def _failing_worker(rank, world, port, stage, bad_rank, out_dir):
    """One rank fails on its own — before the count exchange (stage 1: its partition raises)
    or while growing its receive buffer after the plan (stage 2) — and EVERY rank must come
    back with ShardedSortError instead of waiting inside the all-to-all; the sorter works
    again afterwards."""
    import torch
    import torch.distributed as dist
    from cl_ops_amd.multigpu import ShardedSorter, ShardedSortError
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        class Ops(NumpyLocalOps):
            fail = False

            def msd_partition(self, src, dst, n, bucket_bits):
                if self.fail and stage == 1 and rank == bad_rank:
                    raise MemoryError("injected: no memory for the partition on rank %d" % rank)
                return super().msd_partition(src, dst, n, bucket_bits)

        class Sorter(ShardedSorter):
            def _grow(self, total, like):
                if self.ops.fail and stage == 2 and rank == bad_rank:
                    raise MemoryError("injected: cannot grow the receive buffer on rank %d" % rank)
                return super()._grow(total, like)

        n = 3000
        rng = np.random.default_rng(5 + rank)
        a = rng.integers(0, 1 << 32, n, dtype=np.uint32)
        if stage == 2:          # everything into bucket `bad_rank`: its receive buffer must grow
            a = (a >> np.uint32(world.bit_length() - 1)) | np.uint32(bad_rank << (32 - (world.bit_length() - 1)))
        local = torch.from_numpy(a.view(np.int32).copy())
        ops = Ops("uint")
        ss = Sorter(ops)
        ops.fail = True
        raised, text = False, ""
        try:
            ss.sort(local, n)
        except ShardedSortError as e:
            raised, text = True, str(e)
        ops.fail = False
        out, m = ss.sort(local, n)                      # the same object, right away
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), a)
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), out.numpy()[:m].view(np.uint32).copy())
        with open(os.path.join(out_dir, "err_%d.txt" % rank), "w") as f:
            f.write("%d\n%s" % (int(raised), text))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,stage,bad_rank", [(2, 1, 1), (2, 2, 0), (8, 1, 5), (8, 2, 3)])
def test_ranks_fail_together(tmp_path, world, stage, bad_rank):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_failing_worker, args=(world, port, stage, bad_rank, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        raised, text = (tmp_path / ("err_%d.txt" % r)).read_text().split("\n", 1)
        assert raised == "1", "rank %d did not fail with the others" % r
        if r == bad_rank:
            assert "injected" in text            # the failing rank tells its own story
        else:
            assert str(bad_rank) in text and "no rank sorted" in text
    ins = [np.load(tmp_path / ("in_%d.npy" % r)) for r in range(world)]
    outs = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(outs), np.sort(np.concatenate(ins)))


def test_slice_plan_matches_a_numpy_model():
    """clo_shard_plan_slice (pure host logic of the C driver): a rank's bucket comes as `subs` sub-buckets, slice j =
    `subs / slices` consecutive ones of every rank, travelling as one all-to-all; what rank r sends to p in slice j is
    what p expects from r, the pieces tile the partitioned shard and the receive buffer without gaps."""
    import ctypes as C
    from cl_ops_amd.api import lib
    rng = np.random.default_rng(3)
    for world, subs, slices in ((2, 128, 4), (8, 32, 8), (4, 64, 2), (8, 32, 1), (1, 256, 4), (2, 4, 4)):
        row = world * subs + 5
        m = rng.integers(0, 1000, (world, row)).astype(np.uint64)
        m[:, 3 % (world * subs)] = 0
        flat = np.ascontiguousarray(m.reshape(-1))
        arr = lambda: (C.c_size_t * world)()      # noqa: E731
        group = subs // slices
        plans = {}
        for r in range(world):
            for j in range(slices):
                sc, so, rc, ro = arr(), arr(), arr(), arr()
                at, tot = C.c_size_t(0), C.c_size_t(0)
                total = lib.clo_shard_plan_slice(flat.ctypes.data_as(C.POINTER(C.c_uint64)), row, world, subs, slices, r, j,
                                                 sc, so, rc, ro, C.byref(at), C.byref(tot))
                plans[r, j] = (list(sc), list(so), list(rc), list(ro), at.value, tot.value, total)
        for r in range(world):
            mine = m[r, :world * subs].reshape(world, subs)
            starts = np.concatenate(([0], np.cumsum(mine.reshape(-1))[:-1])).reshape(world, subs)
            covered = 0
            for j in range(slices):
                sc, so, rc, ro, at, tot, total = plans[r, j]
                assert sc == [int(mine[p, j * group:(j + 1) * group].sum()) for p in range(world)]
                assert so == list(starts[:, j * group])
                assert total == int(m[:, r * subs:(r + 1) * subs].sum())
                assert at == covered and tot == sum(rc)
                assert ro == [at + int(sum(rc[:p])) for p in range(world)]
                covered += tot
                for p in range(world):
                    assert sc[p] == plans[p, j][2][r]      # what r sends to p is what p expects from r
            assert covered == plans[r, 0][6]


class NumpyScanOps:
    """CPU stand-in for HipScanOps with the same contract (uint elements)."""

    def __init__(self, sum_dtype):
        self.sum_dtype = np.dtype(sum_dtype)

    def reduce(self, t, n):
        import torch
        tot = int(t.numpy()[:n].view(np.uint32).sum(dtype=np.uint64))
        return torch.tensor([tot - (1 << 64) if tot >= (1 << 63) else tot], dtype=torch.int64)

    def scan(self, src, dst, n, carry):
        import oracle_lib as O
        a = src.numpy()[:n].view(np.uint32)
        c = (int(carry[0]) & 0xFFFFFFFFFFFFFFFF) if carry is not None else 0
        ex = O.serial_scan(a, self.sum_dtype).astype(np.uint64) + np.uint64(c)
        dst.numpy()[:n].view(self.sum_dtype)[:] = ex.astype(self.sum_dtype)


def _scan_worker(rank, world, port, sum_dtype, sizes, big, out_dir):
    import torch
    import torch.distributed as dist
    from cl_ops_amd.multigpu import ShardedScanner
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n = sizes[rank]
        rng = np.random.default_rng(7 + rank)
        a = rng.integers(0, 2**32 if big else 128, n, dtype=np.uint64).astype(np.uint32)
        local = torch.from_numpy(a.view(np.int32).copy())
        sdt = np.dtype(sum_dtype)
        out = torch.zeros(max(n, 1), dtype=torch.int32 if sdt.itemsize == 4 else torch.int64)
        ShardedScanner(NumpyScanOps(sdt)).scan(local, out, n)
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), a)
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), out.numpy()[:n].view(sdt).copy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sum_dtype,sizes,big", [(2, "uint32", (5000, 3001), False), (2, "uint64", (4096, 4096), True),
                                                       (4, "uint32", (100, 0, 777, 1), True)])
def test_sharded_scan_over_gloo(tmp_path, world, sum_dtype, sizes, big):
    """Uneven (and empty) pieces, wrap-around in the sum type: the concatenation
    of the ranks' outputs is the scan of the concatenated input."""
    import torch.multiprocessing as mp
    import oracle_lib as O
    port = _free_port()
    mp.spawn(_scan_worker, args=(world, port, sum_dtype, sizes, big, str(tmp_path)), nprocs=world, join=True)
    a = np.concatenate([np.load(tmp_path / ("in_%d.npy" % r)) for r in range(world)])
    got = np.concatenate([np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, O.serial_scan(a, np.dtype(sum_dtype)))


def test_scan_carry_wraps_mod_2_64():
    from cl_ops_amd.multigpu import ShardedScanner
    assert ShardedScanner.carry_of([5, 7, 9], 0) == 0
    assert ShardedScanner.carry_of([5, 7, 9], 2) == 12
    assert ShardedScanner.carry_of([-1, 2], 2) == 1                       # (2^64 - 1) + 2 mod 2^64
    assert ShardedScanner.carry_of([(1 << 63) - 1, 1], 2) == -(1 << 63)   # bit pattern 0x8000...


def test_exchange_plan_is_consistent():
    from cl_ops_amd.multigpu import ShardedSorter
    m = np.array([[5, 0, 7, 1], [2, 2, 2, 2], [0, 9, 0, 0], [4, 4, 4, 4]])
    plans = [ShardedSorter.plan(m, r) for r in range(4)]
    for r in range(4):
        sc, so, rc, ro = plans[r]
        assert list(sc) == list(m[r]) and list(rc) == list(m[:, r])
        assert list(so) == [0] + list(np.cumsum(m[r])[:-1])
        for d in range(4):                 # what r sends to d is what d expects from r
            assert sc[d] == plans[d][2][r]


def test_world_size_must_be_power_of_two():
    from cl_ops_amd.multigpu import _log2_exact
    assert [_log2_exact(g) for g in (1, 2, 4, 8)] == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        _log2_exact(6)
