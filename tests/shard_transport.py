"""Test infrastructure: a CloShardTransport that moves the bytes of the two exchanges through
host memory over gloo. RCCL refuses two ranks on ONE device, and the GPU box has one: with this
transport the C driver (cl_ops_amd/csrc/clo_shard.c) is the code under test, piece by piece —
any send / receive offsets (the slices of a sliced sort start anywhere in the buffers)."""
import numpy as np


def gloo_staged_transport(rank, world, own_memory=False):
    """own_memory: the transport also provides the receive buffers (CloShardTransport.recv_alloc / recv_free) —
    torch allocations here — and refuses to while `t.alloc_state["fail"]` is set: what running out of memory on ONE
    rank looks like to the sharded sort (its ranks must then fail together)."""
    import torch
    import torch.distributed as dist
    import cl_ops_amd as clo
    from cl_ops_amd import _hip
    from cl_ops_amd._hip import lib

    def d2h(ptr, nbytes, stream):
        h = np.empty(nbytes, dtype=np.uint8)
        _hip.check(lib.clo_hip_memcpy_d2h_async(h.ctypes.data, ptr, nbytes, stream))
        _hip.check(lib.clo_hip_stream_synchronize(stream))
        return h

    def h2d(ptr, h, stream):
        h = np.ascontiguousarray(h)
        _hip.check(lib.clo_hip_memcpy_h2d_async(ptr, h.ctypes.data, h.nbytes, stream))
        _hip.check(lib.clo_hip_stream_synchronize(stream))

    def all_gather(send, recv, count, stream):
        mine = torch.from_numpy(d2h(send, 8 * count, stream))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        h2d(recv, torch.cat(parts).numpy(), stream)
        return 0

    def all_to_all_v(send, sb, so, recv, rb, ro, stream):
        if sb[rank] != rb[rank]:
            return -1
        if sb[rank]:
            h2d(recv + ro[rank], d2h(send + so[rank], sb[rank], stream), stream)
        ops, keep = [], []
        for k in range(1, world):
            dst, src = (rank + k) % world, (rank - k) % world
            if sb[dst]:
                t = torch.from_numpy(d2h(send + so[dst], sb[dst], stream))
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, dst))
            if rb[src]:
                t = torch.empty(rb[src], dtype=torch.uint8)
                keep.append((t, src))
                ops.append(dist.P2POp(dist.irecv, t, src))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for x in keep:
            if isinstance(x, tuple):
                h2d(recv + ro[x[1]], x[0].numpy(), stream)
        return 0

    allocs, state = {}, {"fail": False}

    def recv_alloc(nbytes):
        if state["fail"]:
            return None
        t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device="cuda")
        allocs[t.data_ptr()] = t
        return t.data_ptr()

    def recv_free(ptr):
        allocs.pop(ptr, None)

    tr = clo.ShardTransport.custom(rank, world, all_gather, all_to_all_v, recv_alloc if own_memory else None,
                                   recv_free if own_memory else None)
    tr.alloc_state = state
    return tr
