"""Sorting an array sharded over the GPUs of one node: most-significant-digit
bucket exchange + local radix sorts (SURVEY.md §8e — new functionality, the
reference is single-device).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on
ROCm). Rank r ends up holding bucket r (the keys whose top log2(G) bits equal
r), sorted; concatenating the ranks' results in rank order is the globally
sorted array. Per sort:

  1. local stable partition into G contiguous buckets by the top log2(G) key
     bits; the bucket sizes fall out of it              (HIP kernels, 2 reads + 1 write)
  2. all-gather of the G counts                          (64 B per rank)
  3. all-to-all(v): ONE batch of G-1 send/recv pairs     (RCCL grouped ncclSend/ncclRecv;
     every pair has its own xGMI link, so the G-1 transfers run concurrently)
  4. local satradix sort of the received bucket          (clo_sort_* C API)

torch is plumbing here: it owns the device buffers that RCCL needs and the
process group. All computation goes through the C-ABI of libcl_ops_hip.so on
the tensors' device pointers. The device-side steps are behind the small
`LocalOps` interface so that the host logic (splits, offsets, exchange order)
can be exercised by world_size-2 gloo tests on CPU tensors (tests inject an
oracle-backed LocalOps; the product only ever uses HipLocalOps).
"""
import ctypes as C

import numpy as np


def _log2_exact(g):
    b = g.bit_length() - 1
    if g < 1 or (1 << b) != g:
        raise ValueError("world size must be a power of two, got %d" % g)
    return b


class HipLocalOps:
    """Device-side steps on torch CUDA tensors through the C-ABI."""

    def __init__(self, elem_type, device_index, key_bits=None):
        import torch
        import cl_ops_amd as clo
        from cl_ops_amd import _hip
        self.torch, self.clo, self._hip = torch, clo, _hip
        self.lib = _hip.lib
        if elem_type not in ("uint", "ulong"):
            raise ValueError("the sharded sort handles unsigned 4- and 8-byte keys ('uint', 'ulong'), not %r" % (elem_type,))
        self.elem_type = elem_type
        self.elem_size = 4 if elem_type == "uint" else 8
        self.key_bits = key_bits or 8 * self.elem_size
        self.ctx = clo.Context(device_index)
        self.sorter = clo.Sorter("satradix", self.ctx, elem_type)
        self._queues = {}
        self._ws = None

    # Everything runs on torch's CURRENT stream at the time of the call, so that
    # ordering with RCCL and with the caller's own work is the stream's.
    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    @property
    def queue(self):
        s = self.stream
        q = self._queues.get(s)
        if q is None:
            q = self._queues[s] = self.clo.Queue(self.ctx, stream=s)
        return q

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = self.torch.empty(nbytes, dtype=self.torch.uint8, device="cuda")
        return self._ws

    def msd_histogram(self, t, n, bucket_bits):
        counts = self.torch.empty(1 << bucket_bits, dtype=self.torch.int64, device=t.device)
        self._hip.check(self.lib.clo_hip_msd_histogram(t.data_ptr(), n, self.elem_size, 0, self.key_bits,
                                                       bucket_bits, counts.data_ptr(), self.stream),
                        "clo_hip_msd_histogram")
        return counts

    def msd_partition(self, src, dst, n, bucket_bits):
        """Stable bucket split src -> dst; returns the bucket sizes (int64 tensor on the device)."""
        need = self.lib.clo_hip_msd_workspace_bytes(n, self.elem_size, bucket_bits)
        ws = self._workspace(need)
        counts = self.torch.empty(1 << bucket_bits, dtype=self.torch.int64, device=src.device)
        self._hip.check(self.lib.clo_hip_msd_partition(src.data_ptr(), dst.data_ptr(), n, self.elem_size, 0,
                                                       self.key_bits, bucket_bits, counts.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), self.stream),
                        "clo_hip_msd_partition")
        return counts

    def sort_inplace(self, t, n):
        buf = self.clo.Buffer(self.ctx, n * self.elem_size, device_ptr=t.data_ptr())
        try:
            self.sorter.with_device_data(self.queue, buf, None, n)
        finally:
            buf.close()

    def check(self):
        """Synchronises the streams this object has sorted on and raises CloError(CLO_ERROR_LIBRARY) if a kernel of
        a local sort gave up a bounded look-back spin (small buckets take the single-sweep passes): a caller that only
        ever synchronises through torch would not find out."""
        for q in self._queues.values():
            q.finish()

    def close(self):
        self.sorter.close()
        for q in self._queues.values():
            q.close()
        self._queues = {}
        self.ctx.close()


class ShardedSorter:
    """MSD bucket exchange + local sort over a torch.distributed process group."""

    def __init__(self, ops, group=None, capacity_factor=1.25):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket_bits = _log2_exact(self.world)
        self.capacity_factor = capacity_factor
        self._send = None
        self._recv = None
        # Set to {} to have the following sorts time their phases (device events on
        # the current stream for CUDA tensors, the host clock for CPU tensors);
        # collect_phase_times() then returns the seconds spent per phase so far.
        self.phase_times = None
        self._marks = []

    PHASES = ("partition", "count_exchange", "key_exchange", "local_sort")

    def _mark(self, like):
        if self.phase_times is None:
            return
        if like.is_cuda:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append(e)
        else:
            import time
            self._marks.append(time.perf_counter())

    def collect_phase_times(self):
        """Seconds per phase summed over the sorts made since phase_times was set
        (call after synchronising the device)."""
        out = dict(self.phase_times or {})
        k = len(self.PHASES) + 1
        for i in range(0, len(self._marks) - k + 1, k):
            m = self._marks[i:i + k]
            for j, name in enumerate(self.PHASES):
                dt = m[j].elapsed_time(m[j + 1]) * 1e-3 if hasattr(m[j], "elapsed_time") else m[j + 1] - m[j]
                out[name] = out.get(name, 0.0) + dt
        self._marks = []
        if self.phase_times is not None:
            self.phase_times = out
        return out

    def _buffers(self, like, n):
        cap = int(n * self.capacity_factor) + 1024
        if self._send is None or self._send.numel() < n:
            self._send = self.torch.empty(n, dtype=like.dtype, device=like.device)
        if self._recv is None or self._recv.numel() < cap:
            self._recv = self.torch.empty(cap, dtype=like.dtype, device=like.device)
        return self._send, self._recv

    @staticmethod
    def plan(count_matrix, rank):
        """count_matrix[src][bucket] -> (send_counts, send_offsets, recv_counts, recv_offsets)
        for `rank`. Pure host logic (numpy int64)."""
        m = np.asarray(count_matrix, dtype=np.int64)
        send_counts = m[rank].copy()
        send_offsets = np.concatenate(([0], np.cumsum(send_counts)[:-1]))
        recv_counts = m[:, rank].copy()
        recv_offsets = np.concatenate(([0], np.cumsum(recv_counts)[:-1]))
        return send_counts, send_offsets, recv_counts, recv_offsets

    def exchange(self, send, recv, sc, so, rc, ro):
        """All-to-all(v): own bucket by a device copy, the others as ONE batch of
        send/recv pairs (RCCL: grouped ncclSend/ncclRecv, every pair on its own
        xGMI link). Peers are visited in ring order so that all ranks post matching
        operations in a compatible order."""
        dist, r = self.dist, self.rank
        recv[ro[r]:ro[r] + rc[r]].copy_(send[so[r]:so[r] + sc[r]])
        ops = []
        for k in range(1, self.world):
            dst, src = (r + k) % self.world, (r - k) % self.world
            if sc[dst] > 0:
                ops.append(dist.P2POp(dist.isend, send[so[dst]:so[dst] + sc[dst]], dst, self.group))
            if rc[src] > 0:
                ops.append(dist.P2POp(dist.irecv, recv[ro[src]:ro[src] + rc[src]], src, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def _grow(self, total, like):
        """The receive buffer for `total` keys (more skew than capacity_factor allows for): count-exact."""
        self._recv = self.torch.empty(total, dtype=like.dtype, device=like.device)
        return self._recv

    def sort(self, local, n=None):
        """Sorts the global array whose shards are `local[:n]` on each rank.
        Returns (tensor, m): this rank's bucket, sorted, in tensor[:m].

        COLLECTIVE, and the ranks fail together (the protocol of include/clo_shard.h): a rank
        whose own preparation fails still joins the count exchange and reports it there, so
        that every rank raises ShardedSortError instead of one raising and the others waiting
        inside the all-to-all for ever; growing a receive buffer, the one local step after the
        plan, is followed by a second agreement round."""
        torch, dist = self.torch, self.dist
        n = local.numel() if n is None else n
        if self.world == 1:
            self.ops.sort_inplace(local, n)
            return local, n

        b = self.bucket_bits
        G = self.world
        failure = None
        send = recv = counts = None
        self._mark(local)
        try:
            send, recv = self._buffers(local, n)
            counts = self.ops.msd_partition(local, send, n, b)            # step 1
        except Exception as e:                                            # noqa: BLE001 (reported through the exchange)
            failure = e
        self._mark(local)
        # step 2: counts + [status, receive capacity]; EVERY rank joins, whatever happened above
        row = torch.zeros(G + 2, dtype=torch.int64, device=local.device)
        if failure is None:
            row[:G] = counts
            row[G + 1] = recv.numel()
        else:
            row[G] = 1
        gathered = [torch.empty_like(row) for _ in range(G)]
        dist.all_gather(gathered, row, group=self.group)
        # the host needs the sizes to slice the send / receive buffers: the one
        # device -> host round trip of a sort (G * (G + 2) words)
        full = torch.stack(gathered).cpu().numpy()
        bad = [int(r) for r in range(G) if full[r, G] != 0]
        if bad:
            if failure is not None:
                raise ShardedSortError("rank %d failed before the exchange: %r" % (self.rank, failure)) from failure
            raise ShardedSortError("rank(s) %s failed before the exchange: no rank sorted" % bad)
        matrix = full[:, :G]
        sc, so, rc, ro = self.plan(matrix, self.rank)
        total = int(rc.sum())
        totals = matrix.sum(axis=0)
        if bool(np.any(totals > full[:, G + 1])):                         # somebody has to grow: decided alike everywhere
            ok = 1
            if total > recv.numel():
                try:
                    recv = self._grow(total, local)
                except Exception as e:                                    # noqa: BLE001
                    failure, ok = e, 0
            flag = torch.tensor([ok], dtype=torch.int64, device=local.device)
            flags = [torch.empty_like(flag) for _ in range(G)]
            dist.all_gather(flags, flag, group=self.group)
            bad = [r for r in range(G) if int(flags[r].item()) == 0]
            if bad:
                if failure is not None:
                    raise ShardedSortError("rank %d could not grow its receive buffer: %r" % (self.rank, failure)) from failure
                raise ShardedSortError("rank(s) %s could not grow their receive buffers: no rank sorted" % bad)
        self._mark(local)

        self.exchange(send, recv, sc, so, rc, ro)                         # step 3
        self._mark(local)
        if total > 0:
            self.ops.sort_inplace(recv, total)                            # step 4
        self._mark(local)
        self.last_exchange_bytes = int((sc.sum() - sc[self.rank]) * local.element_size())
        return recv, total


class ShardedSortError(RuntimeError):
    """A sharded sort that no rank carried out because at least one rank could not."""


# ---------------------------------------------------------------------------
# Sharded exclusive scan (new functionality): rank r holds the
# r-th contiguous piece of the array. Per scan:
#   1. local sum of the piece                      (HIP reduce, one read)
#   2. all-gather of the G sums                     (8 B per rank)
#   3. local scan with carry-in = sum of the earlier ranks' sums
#                                                   (HIP scan, one read + one write)
# Three element streams per rank instead of four for scan-then-add. The only
# exchange is the 8-byte all-gather; ranks never wait for each other's scan.
# ---------------------------------------------------------------------------

_SCAN_NP = {"char": (1, 1), "uchar": (1, 0), "short": (2, 1), "ushort": (2, 0), "int": (4, 1), "uint": (4, 0),
            "long": (8, 1), "ulong": (8, 0)}


class HipScanOps:
    """Device-side steps of the sharded scan on torch CUDA tensors through the C-ABI."""

    def __init__(self, elem_type, sum_type, device_index):
        import torch
        from cl_ops_amd import _hip
        self.torch, self._hip, self.lib = torch, _hip, _hip.lib
        self.elem_size, self.elem_signed = _SCAN_NP[elem_type]
        self.sum_size = _SCAN_NP[sum_type][0]
        _hip.check(self.lib.clo_hip_set_device(device_index), "hipSetDevice")
        self._ws = None

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def reduce(self, t, n):
        """Sum of t[:n] mod 2^64 as a 1-element int64 tensor on the device."""
        total = self.torch.empty(1, dtype=self.torch.int64, device=t.device)
        self._hip.check(self.lib.clo_hip_reduce_sum(t.data_ptr(), n, self.elem_size, self.elem_signed,
                                                    total.data_ptr(), self.stream), "clo_hip_reduce_sum")
        return total

    def scan(self, src, dst, n, carry):
        """dst[:n] = carry + exclusive scan of src[:n]; carry: 1-element int64 device tensor or None."""
        need = self.lib.clo_hip_scan_workspace_bytes(n, self.elem_size, self.sum_size)
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None:      # the old range goes back to torch's allocator: the library must stop vouching for it
                self.lib.clo_hip_scan_workspace_forget(self._ws.data_ptr())
            self._ws = self.torch.empty(need, dtype=self.torch.uint8, device=src.device)
            self._hip.check(self.lib.clo_hip_scan_workspace_init(self._ws.data_ptr(), need, self.stream), "clo_hip_scan_workspace_init")
        self._hip.check(self.lib.clo_hip_scan_exclusive_carry(
            src.data_ptr(), dst.data_ptr(), n, self.elem_size, self.elem_signed, self.sum_size,
            carry.data_ptr() if carry is not None else None, None,
            self._ws.data_ptr(), self._ws.numel(), self.stream), "clo_hip_scan_exclusive_carry")

    def check(self):
        """Synchronises the stream and raises HipError(CLO_HIP_ETIMEOUT) if a scan gave up a spin."""
        if self._ws is not None:
            st = self.lib.clo_hip_check_status(self._ws.data_ptr(), self.stream)
            if st != 0:
                self.lib.clo_hip_scan_workspace_init(self._ws.data_ptr(), self._ws.numel(), self.stream)
            self._hip.check(st, "sharded scan")


class ShardedScanner:
    """Exclusive scan of an array whose r-th contiguous piece lives on rank r."""

    def __init__(self, ops, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @staticmethod
    def carry_of(totals, rank):
        """Sum mod 2^64 of the totals of the ranks before `rank`, as a signed 64-bit value."""
        c = sum(int(x) & 0xFFFFFFFFFFFFFFFF for x in totals[:rank]) & 0xFFFFFFFFFFFFFFFF
        return c - (1 << 64) if c >= (1 << 63) else c

    def scan(self, local, out, n=None):
        """out[:n] = exclusive scan of the global array restricted to this rank's piece."""
        torch, dist = self.torch, self.dist
        n = local.numel() if n is None else n
        if self.world == 1:
            self.ops.scan(local, out, n, None)
            return out
        total = self.ops.reduce(local, n)                                   # step 1
        if dist.get_backend(self.group) != "nccl":
            total = total.cpu()          # (gloo moves host tensors; RCCL needs them on the device)
        gathered = [torch.empty_like(total) for _ in range(self.world)]
        dist.all_gather(gathered, total, group=self.group)                  # step 2
        totals = torch.cat(gathered).cpu().tolist()
        carry = torch.tensor([self.carry_of(totals, self.rank)], dtype=torch.int64, device=local.device)
        self.ops.scan(local, out, n, carry)                                 # step 3
        return out

    def check(self):
        """After the caller has synchronised: raises if a scan kernel gave up a
        look-back spin (the output would be wrong)."""
        if hasattr(self.ops, "check"):
            self.ops.check()


# ---------------------------------------------------------------------------
# The same sharded sort behind the C API (include/clo_shard.h): partition, count
# exchange, RCCL all-to-all(v) and local sort are driven by C (clo_shard.c); Python
# only hands over device pointers and, once per process group, carries RCCL's
# 128-byte id from rank 0 to the others.
# ---------------------------------------------------------------------------

class _DeviceView:
    """Device memory owned by the C object, as something torch.as_tensor() can wrap without a copy."""

    def __init__(self, ptr, count, elem_size):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i%d" % elem_size, "data": (ptr, False), "version": 2}


class CShardedSorter:
    """Interface of ShardedSorter (sort(local, n) -> (tensor, m)), implementation in C over RCCL.
    `transport`: a cl_ops_amd.ShardTransport to use instead of RCCL (tests)."""

    PHASES = ShardedSorter.PHASES

    def __init__(self, elem_type, device_index, group=None, options=None, transport=None):
        import torch
        import torch.distributed as dist
        import cl_ops_amd as clo
        self.torch, self.clo = torch, clo
        if elem_type not in ("uint", "ulong"):
            raise ValueError("the sharded sort handles unsigned 4- and 8-byte keys ('uint', 'ulong'), not %r" % (elem_type,))
        self.elem_size = 4 if elem_type == "uint" else 8
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ctx = clo.Context(device_index)
        self._own_transport = transport is None
        if transport is None:
            if self.rank == 0:
                uid = torch.frombuffer(bytearray(clo.ShardTransport.unique_id()), dtype=torch.uint8)
            else:
                uid = torch.zeros(128, dtype=torch.uint8)
            if self.world > 1:
                on_gpu = dist.get_backend(group) == "nccl"
                t = uid.cuda() if on_gpu else uid
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = t.cpu()
            transport = clo.ShardTransport.rccl(bytes(uid.numpy().tobytes()), self.rank, self.world)
        self.transport = transport
        self.ss = clo.ShardSort(self.ctx, transport, elem_type, options)
        self._queues = {}
        self.phase_times = None

    @property
    def queue(self):
        s = self.torch.cuda.current_stream().cuda_stream
        q = self._queues.get(s)
        if q is None:
            q = self._queues[s] = self.clo.Queue(self.ctx, stream=s)
        return q

    def sort(self, local, n=None):
        n = local.numel() if n is None else n
        buf = self.clo.Buffer(self.ctx, max(n, 1) * self.elem_size, device_ptr=local.data_ptr()) if n else None
        try:
            ptr, m = self.ss.with_device_data(self.queue, buf, n)
        finally:
            if buf is not None:
                buf.close()
        if self.phase_times is not None:
            for k, v in self.ss.phase_ms().items():      # (synchronises: phase legs only; zeros when the call had no exchange)
                self.phase_times[k] = self.phase_times.get(k, 0.0) + v * 1e-3
        if m == 0:
            return self.torch.empty(0, dtype=local.dtype, device=local.device), 0
        return self.torch.as_tensor(_DeviceView(ptr, m, self.elem_size), device=local.device).view(local.dtype), m

    def collect_phase_times(self):
        return dict(self.phase_times or {})

    def finish(self, timeout_ms=0):
        """Bounded wait (clo_shard_sort_finish) for what the last sort left running on the current stream's queue: raises
        CloError when `timeout_ms` (0: the `timeout_ms` option; that being 0 too: no bound) have passed or the transport has
        failed — a peer that died must not hang this rank. The object is then only good for close()."""
        self.ss.finish(self.queue, timeout_ms)

    def check(self):
        """ccl_queue_finish on the queues this object has sorted on: synchronises and raises CloError(CLO_ERROR_LIBRARY)
        if a kernel of a local sort gave up a bounded look-back spin (a caller that only synchronises through torch
        would not find out)."""
        for q in self._queues.values():
            q.finish()

    def close(self):
        self.ss.close()
        for q in self._queues.values():
            q.close()
        self._queues = {}
        if self._own_transport:
            self.transport.close()
        self.ctx.close()
