"""bench_c5.py -- bench.py --workload c5: BASELINE configs[4] -- firpfbch2_crcf 256-channel analyzer, output sub-bands sharded over
the ranks (rank r computes k = r + R q from the full input stream), RCCL all-gather over xGMI, assemble
(reference semantics: none in the reference, see DESIGN.md "firpfbch2").

One step = one 2^26-sample block (2^19 steps of 128 inputs -> 256 channels) on EVERY rank's identical input; the block
is fixed as N grows => "scaling": "strong".  Everything runs through the C ABI
(yagi_hip_firpfbch2_crcf_analyzer_execute_sharded_dev, communicator = yagi_hip_comm_*): shard kernel on the object's
stream, all-gather + assemble of chunk k on the communicator's stream beside the shard kernel of chunk k+1.
Reported beside `value`: the shard kernels alone (no exchange), the unchunked serial form, bytes gathered, RCCL's own
rank count.  At N = 1 the call IS the unsharded analyzer (no exchange step exists).
"""
import os
import time

M, SEMI, AS = 256, 4, 60.0
NSAMPLES = 1 << 26
SEED = 0x59414749 + 5
HBM_PEAK_GBS = 8000.0
XGMI_LINK_GBS = 153.0


def run(args, rank, world, dev):
    import torch
    import torch.distributed as dist

    import yagi_amd as ya
    from yagi_amd.dist import Comm

    nsteps = NSAMPLES // (M // 2)
    stream = torch.cuda.current_stream()
    # a rehearsal of the N > 1 path on a box with fewer GPUs than ranks (--dist-backend gloo): RCCL refuses two ranks
    # on one device, so the exchange goes through torch.distributed (host-staged) -- it exercises the shard / gather /
    # assemble orchestration, it is not a measurement of xGMI
    rehearsal = world > 1 and dist.get_backend() != "nccl"
    comm = None if rehearsal else Comm.from_torch_dist(device=dev)
    assert rehearsal or (comm.nranks == world and comm.rank == rank)
    q = ya.FirPfbCh2.new_kaiser(M, SEMI, AS)
    q.set_stream(stream.cuda_stream)
    x = torch.empty(NSAMPLES, dtype=torch.complex64, device=dev)
    y = torch.empty(nsteps * M, dtype=torch.complex64, device=dev)
    ya.gen_complex_dev(SEED, NSAMPLES, out=x, stream=stream.cuda_stream)        # the same stream on every rank
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, reps):
        barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        barrier()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ms = e0.elapsed_time(e1) / reps
        if world > 1:
            t = torch.tensor([ms, wall], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms, wall = float(t[0]), float(t[1])
        return ms, wall

    chunks = args.c5_chunks
    if rehearsal:
        from yagi_amd.dist import firpfbch2_analyze_sharded
        step = lambda: firpfbch2_analyze_sharded(q, x, nsteps, out=y)
    else:
        step = lambda: q.analyzer_execute_sharded_dev(x, nsteps, comm, y, nchunks=chunks)
    if args.prewarm_ms > 0:
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ms, wall = timed(step, args.steps)

    # the same block: shard kernel alone (what a rank computes, no exchange), and the serial unchunked form
    shard = torch.empty(nsteps * (M // world), dtype=torch.complex64, device=dev)
    kern = lambda: q.analyzer_execute_shard_dev(x, nsteps, rank, world, shard)
    kern()
    k_ms, _ = timed(kern, max(3, min(args.steps, 10)))
    if rehearsal:
        s_ms = ms
    else:
        serial = lambda: q.analyzer_execute_sharded_dev(x, nsteps, comm, y, nchunks=1)
        serial()
        s_ms, _ = timed(serial, max(3, min(args.steps, 10)))
    del shard

    parity = None
    if rank == 0:
        try:                                  # last 4 steps vs the CPU oracle fed the tail of the stream
            import numpy as np
            from oracle import oracle
            h = ya.fir_design_kaiser(2 * M * SEMI + 1, 1.0 / M, AS)
            h = (h * (M / h.sum())).astype(np.float32)
            keep, lead = 64, 16                # steps recomputed on the CPU: 16 (even, >= the 15-step history) + 64
            tail = x[NSAMPLES - (keep + lead) * (M // 2):].cpu().numpy()
            qo = oracle.FirPfbCh2(M, SEMI, h[: 2 * M * SEMI])
            ref = qo.analyzer_execute(tail)
            got = y[(nsteps - 4) * M:].cpu().numpy().reshape(4, M)
            parity = float(np.abs(got - ref[-4:]).max() / np.abs(ref[-4:]).max())
        except Exception as e:
            parity = f"unavailable: {e}"

    if rank != 0:
        return None
    step_s = ms / 1e3
    out_bytes_rank = nsteps * (M // world) * 8
    return {
        "metric": "Msamples/sec, firpfbch2_crcf 256-channel analyzer, sub-bands sharded over N MI355X (RCCL all-gather)",
        "value": round(NSAMPLES / step_s / 1e6, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "clock": "HIP events on the object's stream around the K timed steps, MAX over ranks",
        "wall_ms_per_step": round(wall, 4),
        "config": {"workload": f"C5: firpfbch2_crcf analyzer M={M} m={SEMI} As={AS}, one 2^26-sample block "
                               f"({nsteps} steps -> {nsteps * M} channel samples assembled on every rank), identical "
                               "input on every rank, sub-bands k = rank + N q",
                   "parallelism": f"{world} rank(s), sub-band sharding + RCCL all-gather over xGMI" if world > 1 else
                                  "1 rank: the unsharded analyzer (no exchange step)",
                   "chunks": chunks, "rccl_ranks": comm.nranks if comm else 0,
                   "exchange": "torch.distributed gloo, host-staged (REHEARSAL on a shared GPU: not an xGMI number)"
                               if rehearsal else "RCCL ncclAllGather through the C ABI (yagi_hip_comm_*)"},
        "roofline": {"bound": "hbm", "kernel": "firpfbch2_col_kernel (per rank: 8 B in + 16/N B out per input sample)",
                     "kernel_ms": round(k_ms, 4),
                     "achieved": round((8 + 16 / world) * NSAMPLES / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round((8 + 16 / world) * NSAMPLES / k_ms / 1e6 / HBM_PEAK_GBS, 4),
                     "traffic": None},
        "shard_kernel_only": {"ms": round(k_ms, 4), "Msamples_per_s": round(NSAMPLES / k_ms / 1e3, 1)},
        "serial_unchunked": {"ms": round(s_ms, 4), "Msamples_per_s": round(NSAMPLES / s_ms / 1e3, 1)},
        "exchange": {"bytes_sent_per_rank": out_bytes_rank, "bytes_received_per_rank": out_bytes_rank * (world - 1),
                     "xgmi_floor_ms_all_links": round(out_bytes_rank / (XGMI_LINK_GBS * 1e6), 3) if world > 1 else 0.0},
        "parity_max_rel_vs_oracle": parity,
    }
