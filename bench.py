#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:

    Msamples/sec, 256-tap firfilt_crcf + 4096-pt FFT stream, 1/2/4/8 MI355X

One "step" = one pass of the hot path over one batch of synthetic input: the SURVEY.md 8(d) C2 stream,
2^28 complex f32 samples fed as 16 DISTINCT blocks of 2^24 (2 GiB in, 2 GiB of spectra out -- eight times the
256 MiB Infinity Cache, so every block comes from and goes to HBM) through FirFilter<Complex32,f32>::execute_block
semantics (kaiser(256, 0.2, 60 dB) taps, scale 0.4, filter state carried from block to block and step to step)
with every 4096-sample output frame transformed by a forward FFT: 16 calls of
yagi_hip_firfft_crcf_execute_dev, by default its frequency-domain kernel (FFT{h}.FFT{x_f} + FFT{frame-boundary
correction}: one launch per block, the stream crosses HBM once each way).  Inputs and outputs are resident in
HBM when the timed region starts (PCIe excluded).

ONE clock: `value`, `ms_per_step` and `roofline` all come from the HIP events that bracket the K timed steps
on the launch stream (MAX over ranks); the host wall clock over the same region is reported beside it
(`wall_ms_per_step`).

Beside the headline (all untimed w.r.t. `value`):
  l3_resident       the round-1 form of the step (ONE 2^24 block re-run in place: 256 MiB = Infinity-Cache resident)
  direct_form       the same stream through the fused direct-form kernel (MFMA Toeplitz FIR + FFT)
  fast_convolution  the same stream through overlap-save FIR + batched FFT (two launches per block)
  configs           BASELINE configs C2 (default FirFilter kernel and kernel 4), C3 (fft 4096 x 65536), C4
                    (firpfbch 64 ch, 2^26 samples), C5 (firpfbch2 256 ch, 2^26 samples, one GPU, all sub-bands)
  cpu_baseline      the CPU oracle (C restatement of yagi's algorithm, kind "port") on a bounded sample of the same
                    stream: one thread (the reference is single-threaded) and an all-cores time-sharded leg

Multi-GPU: the headline shards into independent streams: each rank filters its own stream, no
data-path collective => "scaling": "weak"; value = all ranks' samples / max time.
`--workload c5` instead runs the one path with an exchange step: the firpfbch2 256-channel analyzer with its
sub-bands sharded over the ranks, RCCL all-gather, assemble (strong scaling: the 2^26-sample block is fixed).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NFFT = 4096
TAPS = 256
BLOCK_FRAMES = 4096                 # 2^24 samples per block (one launch)
STREAM_BLOCKS = 16                  # 2^28 samples per step
SEED = 0x59414749 + 2               # seed + config index (C2 stream)
BYTES_PER_SAMPLE = 16               # 8 B read + 8 B spectrum written (fused)
FLOP_PER_SAMPLE = 4 * TAPS + 60     # 1024 FIR + 5 N log2 N / N FFT
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3


# ------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle timed on this host -- it is the checker's code being timed as the
# "what the reference's algorithm does on these cores" number, never part of the product path
# ------------------------------------------------------------------------------------------------
def cpu_baseline(h, scale, budget_s=8.0, threads=None):
    """one thread: the reference's own structure (firfilt.rs:267-278 per sample, then fft/mod.rs:45-48 per frame);
    all cores: the same stream time-sharded into contiguous chunks, each preloaded with its (L-1)-sample halo"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    L = oracle.lib()
    chunk_frames = 512
    x = oracle.gen_complex(SEED, chunk_frames * NFFT)
    oracle.stream_fir_fft(h, scale, x[: 8 * NFFT], NFFT)          # warm caches / page in
    frames, dt = 0, 0.0
    while dt < budget_s and frames < 64 * chunk_frames:
        t0 = time.perf_counter()
        oracle.stream_fir_fft(h, scale, x, NFFT)
        dt += time.perf_counter() - t0
        frames += chunk_frames
    single = frames * NFFT / dt / 1e6
    out = {"value": round(single, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{frames} frames x {NFFT} samples ({frames * NFFT / 1e6:.0f} Msamples) of the same stream in "
                     f"{chunk_frames}-frame chunks, {dt:.1f} s; oracle/yagi_oracle.c yo_stream_fir_fft = "
                     "sequential-sum firfilt_crcf (firfilt.rs:267-278) + f32 radix-4 FFT, gcc -O2 no fast-math",
           "host_cores_available": os.cpu_count()}

    # all-cores leg: T worker threads (ctypes releases the GIL inside the C call), thread i filters frames
    # [i*F, (i+1)*F) of one stream after pushing the 255 samples before its chunk (halo L-1) -- the
    # time-sharding of SURVEY.md section 5; every thread owns its filter object, FFT plan and buffers
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    T = max(1, min(threads or avail, avail))
    per = 512                                                      # frames per thread and pass
    passes = max(1, int(round(single * 1e6 * budget_s / 1.5 / (per * NFFT))))
    halo = TAPS - 1
    xs = oracle.gen_complex(SEED, T * per * NFFT + halo)           # sample -halo .. : the halo of thread 0
    outs = [np.empty(per * NFFT, np.complex64) for _ in range(T)]
    scr = [np.empty(NFFT, np.complex64) for _ in range(T)]
    objs = []
    for i in range(T):
        q = oracle.FirFilter("crcf", h)
        q.set_scale(scale)
        objs.append((q, oracle.FftPlanF32(NFFT)))

    def work(i):
        q, plan = objs[i]
        lo = i * per * NFFT
        chunk = xs[lo + halo: lo + halo + per * NFFT]
        for _ in range(passes):
            q.execute_block(xs[lo: lo + halo])                     # history = the chunk's halo
            L.yo_stream_fir_fft(q.h, plan.h, oracle._p(chunk), per, oracle._p(scr[i]), oracle._p(outs[i]))

    with ThreadPoolExecutor(T) as ex:
        t0 = time.perf_counter()
        list(ex.map(work, range(T)))
        dt_all = time.perf_counter() - t0
    rate = T * per * passes * NFFT / dt_all / 1e6
    out["all_cores"] = {"value": round(rate, 3), "unit": "Msamples/s", "cores": T, "kind": "port",
                        "sample": f"{T} threads x {per} frames x {passes} passes of one stream, contiguous chunks "
                                  f"with a {halo}-sample halo each (time-sharded), {dt_all:.1f} s wall",
                        "speedup_vs_1_thread": round(rate / single, 2)}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process stays GPU-free (no torch.cuda, no HIP call), starts
    one fresh rank process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), lets them write to
    this process's stdout / stderr (rank 0 prints the JSON line) and returns the worst exit code."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    worst, deadline = 0, None
    live = list(procs)
    while live:
        for pr in list(live):
            rc = pr.poll()
            if rc is None:
                continue
            live.remove(pr)
            if rc != 0:
                worst = worst or rc
                deadline = deadline or time.time() + 30.0      # a rank died: the others cannot finish their collectives
        if deadline and time.time() > deadline:
            for pr in live:                                     # exactly the processes started above
                pr.send_signal(signal.SIGTERM)
            time.sleep(5.0)
            for pr in live:
                if pr.poll() is None:
                    pr.kill()
            worst = worst or 1
            break
        time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stream", choices=["stream", "c5"],
                    help="stream = the headline (default); c5 = firpfbch2 256-ch analyzer, sub-bands sharded over "
                         "the ranks with an RCCL all-gather (the one path with an exchange step)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed device work before the W warmup steps so the GPU clock has left idle "
                         "(a cold MI355X runs the first ~50 launches up to 25 %% slower); 0 disables")
    ap.add_argument("--blocks", type=int, default=STREAM_BLOCKS,
                    help="distinct 2^24-sample blocks per step (16 = the 2^28-sample C2 stream; 1 = the "
                         "Infinity-Cache-resident form of round 1)")
    ap.add_argument("--frames", type=int, default=BLOCK_FRAMES, help="frames of 4096 samples per block")
    ap.add_argument("--variant", type=int, default=0,
                    help="0 auto (= 4 at 256 taps), 1 fused direct-form (sliding VALU FIR), 2 fused MFMA Toeplitz "
                         "FIR, 3 fast convolution (overlap-save kernel + batched FFT, two launches), "
                         "4 frequency-domain filter with frame-boundary correction (one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs (configs, other algorithms)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the all-cores CPU leg (0 = every core this process may run on)")
    ap.add_argument("--c5-chunks", type=int, default=8, help="--workload c5: chunks per block (gather of chunk k "
                                                              "overlaps the shard kernel of chunk k+1)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend for the barrier / max-over-ranks reduction (nccl = RCCL; "
                         "gloo lets several ranks share one GPU when rehearsing the N>1 path on a 1-GPU box)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="headline through plain block calls on one stream instead of the pipelined calls "
                         "(yagi_hip_firfft_crcf_set_pipeline: consecutive blocks overlap on two streams of the object)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import yagi_amd as ya

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.dist_backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    if args.workload == "c5":
        import bench_c5
        out = bench_c5.run(args, rank, world, dev)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, reps, stream, warm_ms=80.0):
        """mean device ms of fn() over reps calls (HIP events on the launch stream).  Every leg first keeps the device busy
        with its own kernel for warm_ms: the legs follow host-side work (object set-up, PCIe copies, the parity check) and a
        GPU that has idled for a few ms runs its next ~50 launches up to 25 % slower"""
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < warm_ms:
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    eff_variant = 4 if args.variant == 0 else args.variant       # library's auto choice at 256 taps
    nframes = args.frames
    nb = args.blocks
    n = nframes * NFFT                                # samples per block (one launch)
    ntot = n * nb                                     # samples per step
    h = ya.fir_design_kaiser(TAPS, 0.2, 60.0)        # FirFilter::new_kaiser(256, 0.2, 60, 0)
    scale = 0.4
    x = torch.empty(ntot, dtype=torch.complex64, device=dev)
    y = torch.empty(ntot, dtype=torch.complex64, device=dev)
    stream = torch.cuda.Stream()                      # the launch stream: every event below is recorded on it
    torch.cuda.set_stream(stream)
    q = ya.FirFftStream(h, NFFT)
    q.set_scale(scale)
    q.set_variant(args.variant)
    q.set_stream(stream.cuda_stream)
    pipelined = eff_variant == 4 and not args.no_pipeline
    q.set_pipeline(pipelined)
    # each rank filters its own stream: rank r's samples are draws [r*2^40 + ...) of the generator
    ya.gen_complex_dev(SEED, ntot, out=x, first=rank << 40, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    xp, yp = x.data_ptr(), y.data_ptr()
    blocks = [(xp + 8 * n * b, yp + 8 * n * b) for b in range(nb)]

    def step():
        for xb, yb in blocks:                         # 16 distinct blocks: nothing is re-read from the Infinity Cache
            q.execute_dev(xb, nframes, yb)
        q.join()                                      # pipelined calls: the launch stream waits for the step's blocks

    # clock pre-conditioning (untimed, not part of the W warmup steps): keep the device busy with the same
    # kernel until the DVFS governor has ramped up from idle
    if args.prewarm_ms > 0:
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                   # HIP events on the launch stream: THE clock of this line

    if world > 1:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64,
                         device=dev if args.dist_backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    # N > 1: the one path with a real exchange step (C5: firpfbch2 sub-bands sharded over the ranks, RCCL all-gather,
    # SURVEY 8e) rides along as an untimed extra so the scaling run measures it on real xGMI; `value` is unaffected.
    # It runs HERE, before rank 0's single-rank extras, so no rank waits in a collective while rank 0 is busy elsewhere.
    # A watchdog (armed behind a barrier every rank has reached) ends a collective that never completes: rank 0 still
    # prints the headline line, and every rank leaves with a NON-ZERO code so the driver records the hang.
    c5_result = None
    if world > 1 and not args.no_extras:
        import copy

        def bail():
            if rank == 0:
                step_s = dev_ms / 1e3 / args.steps
                print(json.dumps({"metric": "Msamples/sec, 256-tap firfilt_crcf + 4096-pt FFT stream",
                                  "value": round(ntot * world / step_s / 1e6, 3), "unit": "Msamples/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                  "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                                  "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                  "config": {"workload": "C2 stream (see the complete line of an N = 1 run)"},
                                  "c5_sharded": {"error": "no result within 180 s (watchdog); every rank exits with "
                                                          "code 3"}}), flush=True)
            os._exit(3)

        barrier()
        dog = threading.Timer(180.0, bail)
        dog.daemon = True
        dog.start()
        try:
            import bench_c5
            a5 = copy.copy(args)
            a5.steps, a5.warmup, a5.prewarm_ms = min(args.steps, 5), 2, 0.0
            c5_result = bench_c5.run(a5, rank, world, dev)
        except Exception as e:                       # reported, never fatal for the headline line
            c5_result = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()

    # in-run parity spot check (rank 0): last frame of the last block vs the oracle (f64 FIR + f64 FFT)
    # on the same input; its 255-sample halo is the preceding samples of the stream
    parity = None
    if rank == 0:
        try:
            from oracle import oracle
            f = nb * nframes - 1
            xs = x[f * NFFT - (TAPS - 1): (f + 1) * NFFT].cpu().numpy()
            yref = oracle.fir_block_f64("crcf", h, xs, scale=scale)[-NFFT:]
            truth = np.fft.fft(yref)
            got = y[f * NFFT:(f + 1) * NFFT].cpu().numpy()
            parity = float(np.linalg.norm(got - truth) / np.linalg.norm(truth))
        except Exception as e:            # the checker must never hide a bench result
            parity = f"unavailable: {e}"

    extras = {}
    if rank == 0 and not args.no_extras:
        reps = max(3, min(args.steps, 10))
        # (1) the round-1 form: ONE block re-run in place (256 MiB working set = the Infinity Cache)
        qr = ya.FirFftStream(h, NFFT)
        qr.set_scale(scale)
        qr.set_variant(args.variant)
        qr.set_stream(stream.cuda_stream)
        for _ in range(20):
            qr.execute_dev(xp, nframes, yp)
        r_ms = timed(lambda: qr.execute_dev(xp, nframes, yp), 16 * reps, stream)
        extras["l3_resident"] = {
            "what": "one 2^24-sample block re-run in place (128 MiB in + 128 MiB out = Infinity-Cache resident), "
                    "the round-1 headline; NOT the value of this line",
            "ms_per_block": round(r_ms, 4), "value": round(n / r_ms / 1e3, 1), "unit": "Msamples/s",
            "frac_of_hbm_peak": round(BYTES_PER_SAMPLE * n / r_ms / 1e6 / HBM_PEAK_GBS, 4)}

        # (1b) PCIe-inclusive rate of the headline (never `value`): the host-pointer entry point moves one 2^24-sample
        # block H2D, runs the same kernel, and moves the spectra D2H (pageable numpy buffers)
        try:
            xh = x[:n].cpu().numpy()
            qr.execute(xh)
            t_p = time.perf_counter()
            for _ in range(3):
                qr.execute(xh)
            p_ms = (time.perf_counter() - t_p) / 3 * 1e3
            extras["pcie_inclusive"] = {
                "what": "yagi_hip_firfft_crcf_execute on host buffers: H2D 128 MiB + kernel + D2H 128 MiB per "
                        "2^24-sample block, wall clock; NOT the value of this line",
                "ms_per_block": round(p_ms, 3), "value": round(n / p_ms / 1e3, 1), "unit": "Msamples/s",
                "pcie_GBps": round(16 * n / p_ms / 1e6, 2)}
        except Exception as e:
            extras["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"}

        # (1c) the same kernel through plain block calls on ONE stream: every launch then waits for the complete drain
        # of the one before it, and rocprofv3's mean launch duration IS the launch-to-launch time (with the pipelined
        # calls of `value` about two launches are in flight at a time: a launch lasts ~1.7x the launch-to-launch time)
        if pipelined:
            qp = ya.FirFftStream(h, NFFT)
            qp.set_scale(scale)
            qp.set_variant(args.variant)
            qp.set_stream(stream.cuda_stream)

            def run_plain():
                for xb, yb in blocks:
                    qp.execute_dev(xb, nframes, yb)
            run_plain()
            p_ms = timed(run_plain, reps, stream) / nb
            extras["plain_block_calls"] = {
                "what": "the same stream and kernel, execute_dev without set_pipeline (one stream, launches back to back); "
                        "NOT the value of this line",
                "ms_per_block": round(p_ms, 5), "value": round(n / p_ms / 1e3, 1), "unit": "Msamples/s",
                "frac_of_hbm_peak": round(BYTES_PER_SAMPLE * n / p_ms / 1e6 / HBM_PEAK_GBS, 4)}

        # (2) the same stream through the other algorithms
        def time_variant(v):
            qd = ya.FirFftStream(h, NFFT)
            qd.set_scale(scale)
            qd.set_variant(v)
            qd.set_stream(stream.cuda_stream)

            def run():
                for xb, yb in blocks:
                    qd.execute_dev(xb, nframes, yb)
            run()
            return timed(run, reps, stream) / nb

        if eff_variant == 4:
            d_ms = time_variant(2)        # fused MFMA Toeplitz FIR + FFT: the faster of the two direct forms
            extras["direct_form"] = {
                "value": round(n / d_ms / 1e3, 1), "unit": "Msamples/s per GPU", "ms_per_block": round(d_ms, 4),
                "kernel": "fir_crcf_mfma_kernel<68, true> (fused direct-form MFMA Toeplitz FIR + FFT, 16 B/sample)",
                "fp32_tflops": round(FLOP_PER_SAMPLE * n / (d_ms / 1e3) / 1e12, 2),
                "frac_of_fp32_peak": round(FLOP_PER_SAMPLE * n / (d_ms / 1e3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
            c_ms = time_variant(3)
            extras["fast_convolution"] = {
                "value": round(n / c_ms / 1e3, 1), "unit": "Msamples/s per GPU", "ms_per_block": round(c_ms, 4),
                "kernel": "firfilt_fftconv_kernel<0> + fft4096_kernel<-1> (overlap-save FIR, then batched FFT; "
                          "32 B/sample)"}

        # (3) BASELINE configs C2..C5 on this GPU (each: mean device ms over `reps` passes after one warm pass,
        # algorithmic GB/s, fraction of the 8 TB/s HBM peak; C2's direct form also as FP32 TFLOP/s)
        cfg = {}

        def leg(name, fn, units, bytes_per_unit, note, flop_per_unit=None):
            fn()
            ms = timed(fn, reps, stream)
            gbs = bytes_per_unit * units / ms / 1e6
            e = {"ms": round(ms, 4), "units": units, "Gunits_per_s": round(units / ms / 1e6, 2),
                 "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "what": note}
            if flop_per_unit:
                tf = flop_per_unit * units / ms / 1e9
                e["fp32_tflops"] = round(tf, 2)
                e["frac_of_fp32_peak"] = round(tf / FP32_PEAK_TFLOPS, 4)
            cfg[name] = e

        for kname, kern, note in (("C2_firfilt_crcf_256_default", 0, "FirFilter::execute_block default kernel "
                                   "(direct form on the matrix pipe: fir_crcf_mfma_stream_kernel<68>, bit-exact on "
                                   "integers), 2^28-sample stream in 2^24 blocks"),
                                  ("C2_firfilt_crcf_256_kernel4", 4, "set_kernel(4): overlap-save fast convolution, "
                                   "same stream")):
            qf = ya.FirFilter("crcf", h)
            qf.set_scale(scale)
            qf.set_kernel(kern)
            qf.set_stream(stream.cuda_stream)
            qf.set_pipeline(not args.no_pipeline)

            def run(qf=qf):
                for xb, yb in blocks:
                    qf.execute_block_dev(xb, n, yb)
                qf.join()
            leg(kname, run, ntot, 16, note + ("; pipelined block calls, joined once per pass" if not args.no_pipeline
                                             else "; plain block calls"),
                flop_per_unit=4 * TAPS if kern == 0 else None)
        # C1 (BASELINE configs[0], the reference's own CPU-runnable case): firfilt_rrrf 63 taps over 1 M real samples.
        # GPU: ONE execute_block_dev call on device-resident data (launch-bound at this size), and the host-pointer
        # call (H2D + kernel + D2H); CPU: the oracle's FirFilter::execute_block on the same samples, one thread.
        try:
            from oracle import oracle
            n1 = 1 << 20
            h1 = ya.fir_design_kaiser(63, 0.2, 60.0)
            x1h = oracle.gen_real(SEED - 2, n1)
            x1 = torch.from_numpy(x1h).to(dev)
            y1 = torch.empty_like(x1)
            q1 = ya.FirFilter("rrrf", h1)
            q1.set_scale(0.4)
            q1.set_stream(stream.cuda_stream)
            f1 = lambda: q1.execute_block_dev(x1.data_ptr(), n1, y1.data_ptr())
            f1()
            g_ms = timed(f1, 50, stream)
            torch.cuda.synchronize()
            q1.execute_block(x1h)
            t_h = time.perf_counter()
            for _ in range(5):
                q1.execute_block(x1h)
            host_ms = (time.perf_counter() - t_h) / 5 * 1e3
            o1 = oracle.FirFilter("rrrf", oracle.fir_design_kaiser(63, 0.2, 60.0))
            o1.set_scale(0.4)
            o1.execute_block(x1h[:1 << 16])
            t_c = time.perf_counter()
            want = o1.execute_block(x1h)
            cpu_ms = (time.perf_counter() - t_c) * 1e3
            del want
            cfg["C1_firfilt_rrrf_63"] = {
                "units": n1, "what": "firfilt_rrrf kaiser(63, 0.2, 60 dB) scale 0.4 over 2^20 real f32 samples, one block",
                "gpu_ms_device_resident": round(g_ms, 5), "gpu_Gunits_per_s": round(n1 / g_ms / 1e6, 2),
                "gpu_note": "one yagi_hip_firfilt_rrrf_execute_block_dev call, launch-to-launch on the stream "
                            "(4 MiB in + 4 MiB out, re-run in place: Infinity-Cache resident)",
                "gpu_ms_host_pointers": round(host_ms, 4),
                "gpu_host_note": "yagi_hip_firfilt_rrrf_execute_block from pageable numpy buffers (H2D + kernel + D2H, "
                                 "PCIe-inclusive wall time)",
                "cpu_oracle_ms": round(cpu_ms, 3), "cpu_Munits_per_s": round(n1 / cpu_ms / 1e3, 2),
                "cpu_note": "oracle/yagi_oracle.c FirFilter::execute_block (firfilt.rs:267-278, sequential sums), "
                            "one thread, gcc -O2 no fast-math"}
        except Exception as e:
            cfg["C1_firfilt_rrrf_63"] = {"error": f"{type(e).__name__}: {e}"}
        plan = ya.Fft(NFFT, ya.Direction.Forward)
        nt = ntot // NFFT
        leg("C3_fft4096_batch", lambda: plan.run_batch_dev(xp, yp, nt, stream.cuda_stream), ntot, 16,
            f"fft 4096-pt forward, batch {nt} (2 GiB in + 2 GiB out at the default 16 blocks), one launch")
        nc = min(1 << 26, ntot // 2)
        c4 = ya.FirPfbCh.new_kaiser(64, 8, 60.0)
        c4.set_stream(stream.cuda_stream)
        leg("C4_firpfbch_64ch", lambda: c4.analyzer_execute_dev(xp, nc // 64, yp), nc, 16,
            f"firpfbch_crcf analyzer M=64 m=8 As=60, {nc} input samples, one launch")
        c5 = ya.FirPfbCh2.new_kaiser(256, 4, 60.0)
        c5.set_stream(stream.cuda_stream)
        leg("C5_firpfbch2_256ch_1gpu", lambda: c5.analyzer_execute_dev(xp, nc // 128, yp), nc, 24,
            f"firpfbch2_crcf analyzer M=256 m=4 As=60, {nc} input samples -> {2 * nc} outputs, all sub-bands on "
            "one GPU (the sharded form: bench.py --workload c5)")
        r2 = ya.Resamp2.new("crcf", 12, 0.0, 60.0)
        r2.set_stream(stream.cuda_stream)
        leg("F4_resamp2_crcf_decim", lambda: r2.execute_block_dev(ya.Resamp2.DECIM, xp, nc, yp), nc, 12,
            f"Resamp2<Complex32,f32> m=12 (24-tap half-band branch + delay), decim_execute over {nc} samples "
            "(SURVEY 8f-4): 8 B in + 4 B out per input sample, 2 launches")
        ms2 = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 3, 0.4, 0.0, 60.0)
        ms2.set_stream(stream.cuda_stream)
        leg("F4_msresamp2_crcf_decim8", lambda: ms2.execute_block_dev(xp, nc // 8, yp), nc, 9,
            f"MsResamp2 decimator by 8 (3 half-band stages, {ms2.get_stage_lengths()}), {nc} input samples; "
            "algorithmic bytes 8 in + 1 out per input sample; one launch, the stage intermediates stay in LDS")
        extras["configs"] = cfg

    kernel_name = {1: "firfft_crcf_4096_slide_kernel", 2: "fir_crcf_mfma_kernel<68, true>",
                   3: "firfilt_fftconv_kernel<0> + fft4096_kernel<-1> (two launches per block)",
                   4: "firfft_crcf_4096_freq_kernel"}[eff_variant]
    # algorithmic bytes per input sample: fused = 8 in + 8 out; the two-kernel fast-convolution form also
    # writes and re-reads the FIR output stream ("32 if run as two kernels -- state which")
    bytes_per_sample = 32 if eff_variant == 3 else BYTES_PER_SAMPLE
    # executed flops per input sample: direct 4*L + FFT 60; fast convolution 2 FFTs per 3841 outputs + product + FFT
    # frequency-domain form: one FFT (60) + 2/3 of one for the zero-padded correction (~42) + the triangular
    # correction sum 4*(255*256/2)/4096 (32) + the FFT{h} product, scale and final add (10)
    flop_per_sample = {3: round(2 * 60 * 4096 / 3841 + 6 + 60), 4: 144}.get(eff_variant, FLOP_PER_SAMPLE)
    traffic = None          # HBM bytes per launch from the committed PMC passes (profiles/traffic.json)
    try:
        tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        if tj.get("samples_per_launch") == n:
            tv = tj["variants"][str(eff_variant)]
            traffic = tv.get("dominant_kernel_hbm_bytes", tv["hbm_bytes"])
    except Exception:
        traffic = None

    if rank == 0:
        launches = args.steps * nb
        step_s = dev_ms / 1e3 / args.steps           # ONE clock: HIP events around the K timed steps
        kern_s = dev_ms / 1e3 / launches             # average launch-to-launch time of one 2^24-sample block
        value = ntot * world / step_s / 1e6
        achieved = bytes_per_sample * n / kern_s / 1e9
        out = {
            "metric": "Msamples/sec, 256-tap firfilt_crcf + 4096-pt FFT stream",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "clock": "HIP events on the launch stream around the K timed steps (value, ms_per_step, roofline); "
                     "host wall clock over the same region in wall_ms_per_step",
            "wall_ms_per_step": round(wall / args.steps * 1e3, 4),
            "config": {"workload": f"C2 stream: 2^{int(np.log2(ntot))} complex f32 samples per step as {nb} distinct "
                                   f"2^{int(np.log2(n))}-sample blocks ({8 * ntot >> 20} MiB in + {8 * ntot >> 20} MiB "
                                   "out, HBM-resident, beyond the 256 MiB Infinity Cache), firfilt_crcf 256-tap "
                                   "(kaiser 0.2/60dB, scale 0.4, state carried) -> 4096-pt forward FFT per frame "
                                   "(BASELINE configs[1] feeding configs[2])",
                       "algorithm": {3: "fast convolution (overlap-save, 4096-pt blocks) + batched FFT",
                                     4: "frequency-domain filter: FFT{h}.FFT{frame} + FFT{frame-boundary correction}"
                                     }.get(eff_variant, "fused direct-form FIR + FFT"),
                       "samples_per_step_per_gpu": ntot, "blocks_per_step": nb, "samples_per_block": n,
                       "frames_per_block": nframes, "nfft": NFFT, "taps": TAPS,
                       "parallelism": f"{world} independent stream(s), no collective",
                       "block_calls": "pipelined (set_pipeline: consecutive execute_dev calls alternate between two "
                                      "streams of the object, joined into the launch stream once per step)"
                                      if pipelined else "plain (one stream)",
                       "kernel": kernel_name, "variant": args.variant, "prewarm_ms": args.prewarm_ms},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, bytes per launch; "
                                           "profiles/traffic.json" if traffic else None,
                         "kernel": kernel_name.split(" + ")[0],
                         "kernel_ms": round(kern_s * 1e3, 5),
                         "launches_timed": launches,
                         "algorithmic_bytes_per_launch": int(bytes_per_sample * n),
                         "note": {3: "fast convolution: overlap-save FIR kernel + batched FFT, FIR output stream "
                                     "crosses HBM once; kernel_ms = both launches of a block",
                                  4: "one launch per 2^24-sample block; the stream is read once and the spectra "
                                     "written once; kernel_ms = timed region / launches = launch-to-launch time.  With "
                                     "the pipelined block calls about two launches are in flight at a time (each lasts ~1.7x "
                                     "kernel_ms in a rocprofv3 trace; profiles/r03_bench_span.txt derives the same "
                                     "launch-to-launch time from the trace); `plain_block_calls` is the one-stream form"
                                  }.get(eff_variant, "direct-form 256-tap crcf is FP32-ALU bound (64 flop/B); see fp32"),
                         "fp32": {"achieved_tflops": round(flop_per_sample * n / kern_s / 1e12, 2),
                                  "peak_tflops": FP32_PEAK_TFLOPS, "flop_per_sample": flop_per_sample,
                                  "frac": round(flop_per_sample * n / kern_s / 1e12 / FP32_PEAK_TFLOPS, 4)}},
            "parity_rel_l2_vs_f64": parity,
        }
        out.update(extras)
    if rank == 0 and c5_result is not None:
        out["c5_sharded"] = c5_result
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            out["cpu_baseline"] = cpu_baseline(oracle.fir_design_kaiser(TAPS, 0.2, 60.0), scale,
                                               threads=args.cpu_threads or None)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
