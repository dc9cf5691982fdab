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


def all_gather_subbands(shard, group=None):
    """all-gather equal-size shard tensors into one [world * shard.numel()] tensor (rank-major)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty(world * shard.numel(), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(out, shard.reshape(-1).contiguous(), group=group)
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
