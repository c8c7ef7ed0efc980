"""CPU stand-ins for the device-side steps of cl_ops_amd/multigpu.py (same contract as
HipLocalOps / HipScanOps), numpy / oracle backed. Test infrastructure: injected by the gloo
tests and by `bench.py --dry-run` (a launcher test); the product only ever uses the HIP ops."""
import numpy as np


class NumpyLocalOps:
    """CPU stand-in for HipLocalOps with the same contract."""

    def __init__(self, elem_type):
        self.np_dtype = np.uint32 if elem_type == "uint" else np.uint64
        self.key_bits = 32 if elem_type == "uint" else 64

    def _view(self, t, n):
        return t.numpy()[:n].view(self.np_dtype)

    def msd_histogram(self, t, n, bucket_bits):
        import torch
        b = self._view(t, n) >> self.np_dtype(self.key_bits - bucket_bits)
        return torch.from_numpy(np.bincount(b.astype(np.int64), minlength=1 << bucket_bits).astype(np.int64))

    def msd_partition(self, src, dst, n, bucket_bits):
        import torch
        a = self._view(src, n)
        b = a >> self.np_dtype(self.key_bits - bucket_bits)
        self._view(dst, n)[:] = a[np.argsort(b, kind="stable")]
        return torch.from_numpy(np.bincount(b.astype(np.int64), minlength=1 << bucket_bits).astype(np.int64))

    def sort_inplace(self, t, n):
        import oracle_lib as O
        v = self._view(t, n)
        v[:] = O.stable_sort(v.copy())


def host_staged_sorter_class():
    """ShardedSorter whose all-to-all goes through host memory over gloo: for tests that put
    two ranks on ONE GPU (RCCL refuses two ranks on one device). Everything else — partition,
    count exchange plan, local sort — is the product path."""
    import torch
    import torch.distributed as dist
    from cl_ops_amd.multigpu import ShardedSorter

    class HostStaged(ShardedSorter):
        def exchange(self, send, recv, sc, so, rc, ro):
            hs, hr = send.cpu(), torch.empty(recv.numel(), dtype=recv.dtype)
            r = self.rank
            hr[ro[r]:ro[r] + rc[r]] = hs[so[r]:so[r] + sc[r]]
            ops = []
            for k in range(1, self.world):
                dst, src = (r + k) % self.world, (r - k) % self.world
                if sc[dst] > 0:
                    ops.append(dist.P2POp(dist.isend, hs[so[dst]:so[dst] + sc[dst]].contiguous(), dst))
                if rc[src] > 0:
                    ops.append(dist.P2POp(dist.irecv, hr[ro[src]:ro[src] + rc[src]], src))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            recv.copy_(hr)

    return HostStaged
