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
        if skew:   # most keys in one bucket: exercises the receive-capacity path
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
                                                     (8, "uint", 1500, False), (8, "ulong", 700, True)])   # the driver's N = 8 plan
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
