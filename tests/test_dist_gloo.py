"""world_size-2 (and 4) gloo tests, CPU only: the partitioning and gather/assemble orchestration of
the one path that has an exchange step (firpfbch2 sub-band sharding), and the
stream partition of the replicated benchmark.  The per-rank shard data comes from the oracle, the
collective is torch.distributed (gloo here, nccl = RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, m, nsteps, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle
        from yagi_amd import dist as yd
        h = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
        h = (h * M / h.sum()).astype(np.float32)
        x = oracle.gen_complex(0x59414749 + 5, nsteps * (M // 2))      # every rank reads the full input
        full = oracle.FirPfbCh2(M, m, h).analyzer_execute(x)            # [nsteps, M]
        mine = full[:, yd.subband_indices(rank, world, M)]              # this rank's [step][M/R] slab
        gathered = yd.all_gather_subbands(torch.from_numpy(np.ascontiguousarray(mine)))
        y = gathered.numpy().reshape(-1)[yd.gathered_index_map(nsteps, M, world)]
        ok = np.array_equal(y, full)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t[0]) == float(world)
        offs = [yd.stream_offset(r, world) for r in range(world)]
        ok = ok and len(set(offs)) == world and all(b - a >= 1 << 40 for a, b in zip(offs, offs[1:]))
        np.save(os.path.join(result_dir, f"ok{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M", [(2, 256), (4, 64), (2, 8)])
def test_subband_sharding_gather_assemble(tmp_path, world, M):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, M, 2, 24, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"ok{r}.npy")[0], f"rank {r}"


def test_partition_helpers():
    from yagi_amd import dist as yd
    assert list(yd.subband_indices(3, 8, 256)[:3]) == [3, 11, 19]
    allk = np.sort(np.concatenate([yd.subband_indices(r, 8, 256) for r in range(8)]))
    assert np.array_equal(allk, np.arange(256))
    with pytest.raises(ValueError):
        yd.subband_indices(0, 3, 256)
    m = yd.gathered_index_map(5, 12, 3)
    assert sorted(m.reshape(-1)) == list(range(60))
