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
