"""Multi-GPU host orchestration (one process per GPU, torch.distributed; backend "nccl" = RCCL).

Only the firpfbch2 analyzer has an exchange step: rank r computes the
sub-bands k = r + R*q from the full input stream, the ranks all-gather their [step][M/R] slabs,
and a permutation kernel assembles [step][channel].  Everything else on the hot path shards into
independent streams with no collective (see bench.py).

The index helpers are pure integer functions so the partitioning can be tested on CPU (gloo).
"""
import numpy as np


def subband_indices(rank, nranks, M):
    """channels owned by `rank`: k = rank + nranks*q, q < M/nranks (decimation in frequency)"""
    if nranks < 1 or M % nranks or not 0 <= rank < nranks:
        raise ValueError(f"{M} channels do not shard over {nranks} ranks")
    return rank + nranks * np.arange(M // nranks)


def gathered_index_map(nsteps, M, nranks):
    """flat index into the all-gathered [rank][step][M/R] buffer for every (step, channel):
    y[s, k] = gathered.flat[map[s, k]] -- what yagi_hip_firpfbch2_crcf_assemble_dev computes."""
    Mr = M // nranks
    s = np.arange(nsteps)[:, None]
    k = np.arange(M)[None, :]
    return ((k % nranks) * nsteps + s) * Mr + k // nranks


def stream_offset(rank, nranks):
    """first generator draw of rank's private stream in the replicated benchmarks (bench.py)"""
    return int(rank) << 40


class Comm:
    """RCCL communicator owned through the C ABI (yagi_hip_comm_*): what a Rust / C host would hold.
    `Comm.from_torch_dist()` bootstraps it from an initialised torch.distributed group: rank 0 draws the
    128-byte unique id, the group broadcasts it, every rank joins (the current HIP device is the rank's GPU)."""

    ID_BYTES = 128

    def __init__(self, uid, rank, nranks):
        import ctypes as C
        from . import _check, lib
        buf = (C.c_ubyte * self.ID_BYTES).from_buffer_copy(bytes(uid))
        h = C.c_void_p()
        _check(lib.yagi_hip_comm_create(buf, int(rank), int(nranks), C.byref(h)))
        self._h = h
        r, n = C.c_int(), C.c_int()
        _check(lib.yagi_hip_comm_rank(self._h, C.byref(r), C.byref(n)))
        self.rank, self.nranks = r.value, n.value        # as RCCL reports them (ncclCommUserRank / ncclCommCount)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _check, lib
        buf = (C.c_ubyte * Comm.ID_BYTES)()
        _check(lib.yagi_hip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch_dist(cls, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls(cls.unique_id(), 0, 1)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"
        dev = device if (on_gpu and device is not None) else (torch.device("cuda", torch.cuda.current_device())
                                                              if on_gpu else torch.device("cpu"))
        t = torch.zeros(cls.ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0, group=group)
        return cls(bytes(t.cpu().tolist()), rank, world)

    def all_gather_dev(self, send_dev, recv_dev, bytes_per_rank, stream=None):
        from . import _check, _devptr, lib
        _check(lib.yagi_hip_comm_all_gather_dev(self._h, _devptr(send_dev), _devptr(recv_dev), bytes_per_rank, stream))

    def destroy(self):
        h, self._h = getattr(self, "_h", None), None
        if not h:
            return
        try:
            from . import lib
        except Exception:          # interpreter shutting down: the process exit releases the communicator
            return
        if lib is not None:
            lib.yagi_hip_comm_destroy(h)

    def __del__(self):
        self.destroy()


def all_gather_subbands(shard, group=None):
    """all-gather equal-size shard tensors into one [world * shard.numel()] tensor (rank-major).  Backend nccl (=
    RCCL) gathers device tensors in place; gloo (CPU tests, or several ranks sharing one GPU) goes through the host."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = shard.reshape(-1).contiguous()
    if flat.is_cuda and dist.get_backend(group) != "nccl":
        host = flat.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return torch.cat(parts).to(shard.device)
    out = torch.empty(world * flat.numel(), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out


def firpfbch2_analyze_sharded(q, x, nsteps, group=None, out=None):
    """firpfbch2 analyzer with sub-bands sharded over the ranks of `group` (GPU path).
    q: yagi_amd.FirPfbCh2 (same object on every rank), x: device tensor with nsteps*M/2 complex64.
    Returns a device tensor [nsteps, M] identical on every rank."""
    import torch
    import torch.distributed as dist
    from . import FirPfbCh2
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    M = q.M
    subband_indices(rank, world, M)           # validates the partition
    stream = torch.cuda.current_stream().cuda_stream
    q.set_stream(stream)
    shard = torch.empty(nsteps * (M // world), dtype=torch.complex64, device=x.device)
    q.analyzer_execute_shard_dev(x, nsteps, rank, world, shard)
    gathered = all_gather_subbands(shard, group)          # RCCL all-gather over xGMI
    if out is None:
        out = torch.empty(nsteps * M, dtype=torch.complex64, device=x.device)
    FirPfbCh2.assemble_dev(gathered, nsteps, M, world, out, stream)
    return out.reshape(nsteps, M)
