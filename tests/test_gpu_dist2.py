"""Two real processes on the one GPU of the box: each rank runs its HIP shard kernel (sub-bands k = rank + 2 q), the
slabs are all-gathered between the PROCESSES (gloo: RCCL refuses two ranks on one device; on a node the same call
runs over RCCL/xGMI) and assembled by the HIP permutation kernel -- the shard -> gather -> assemble path with
world_size > 1, which the single-process tests cannot reach."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, m, nsteps, result_dir):
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import yagi_amd as ya
        from oracle import oracle
        from yagi_amd import dist as yd
        torch.cuda.set_device(0)
        x = oracle.gen_complex(0x59414749 + 5, nsteps * (M // 2))           # every rank reads the full input
        q = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
        xt = torch.from_numpy(x).cuda()
        y = yd.firpfbch2_analyze_sharded(q, xt, nsteps)                      # shard kernel, gather, assemble kernel
        torch.cuda.synchronize()
        want = ya.FirPfbCh2.new_kaiser(M, m, 60.0).analyzer_execute(x)
        err = float(np.linalg.norm(y.cpu().numpy() - want) / np.linalg.norm(want))
        np.save(os.path.join(result_dir, f"err{rank}.npy"), np.array([err]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,nsteps", [(256, 4096 + 64), (64, 512)])
def test_two_process_shard_gather_assemble(tmp_path, M, nsteps):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, M, 4, nsteps, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"err{r}.npy")[0] <= 2e-6, f"rank {r}"
