#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:

    Msamples/sec, 256-tap firfilt_crcf + 4096-pt FFT stream, 1/2/4/8 MI355X

One "step" = one pass of the hot path over one batch of synthetic input: a 2^24-sample block
(4096 frames of 4096) of complex f32 through FirFilter<Complex32,f32>::execute_block semantics
(kaiser(256, 0.2, 60 dB) taps, scale 0.4, state carried from step to step) with every
4096-sample output frame transformed by a forward FFT -- yagi_hip_firfft_crcf_execute_dev, by default
its frequency-domain kernel (FFT{h}.FFT{x_f} + FFT{frame-boundary correction}: one launch, the stream
crosses HBM once each way).  The same workload through the direct-form (MFMA Toeplitz FIR + FFT) and
the overlap-save forms is timed beside it ("direct_form", "fast_convolution"), untimed w.r.t. `value`.
Inputs and outputs are resident in HBM (PCIe excluded).

Multi-GPU: the path shards into independent streams: each rank filters its
own stream, no data-path collective => "scaling": "weak"; value = all ranks' samples / max time.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel vs the HBM roofline,
HIP-event timed on the launch stream) and "cpu_baseline" (the CPU oracle -- a C restatement of
yagi's algorithm, `kind: "port"` -- timed on this host on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NFFT = 4096
TAPS = 256
BLOCK_FRAMES = 4096                 # 2^24 samples per step
SEED = 0x59414749 + 2               # seed + config index (C2 stream)
BYTES_PER_SAMPLE = 16               # 8 B read + 8 B spectrum written (fused)
FLOP_PER_SAMPLE = 4 * TAPS + 60     # 1024 FIR + 5 N log2 N / N FFT
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3


def cpu_baseline(h, scale, budget_s=12.0):
    """time the oracle (single thread, C -O2, no fast-math) on a bounded sample of the same stream:
    1024-frame chunks of the C2 stream until ~budget_s of CPU work has been done"""
    from oracle import oracle
    chunk_frames = 1024
    x = oracle.gen_complex(SEED, chunk_frames * NFFT)
    oracle.stream_fir_fft(h, scale, x[: 8 * NFFT], NFFT)          # warm caches / page in
    frames, dt = 0, 0.0
    while dt < budget_s and frames < 64 * chunk_frames:
        t0 = time.perf_counter()
        oracle.stream_fir_fft(h, scale, x, NFFT)
        dt += time.perf_counter() - t0
        frames += chunk_frames
    return {"value": round(frames * NFFT / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{frames} frames x {NFFT} samples ({frames * NFFT / 1e6:.0f} Msamples) of the same stream in "
                      f"{chunk_frames}-frame chunks, {dt:.1f} s; oracle/yagi_oracle.c yo_stream_fir_fft = "
                      "sequential-sum firfilt_crcf (firfilt.rs:267-278) + f32 radix-4 FFT, gcc -O2 no fast-math",
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed device work before the W warmup steps so the GPU clock has left idle "
                         "(a cold MI355X runs the first ~50 launches up to 25 %% slower); 0 disables")
    ap.add_argument("--frames", type=int, default=BLOCK_FRAMES, help="frames of 4096 samples per step")
    ap.add_argument("--variant", type=int, default=0,
                    help="0 auto (= 4 at 256 taps), 1 fused direct-form (sliding VALU FIR), 2 fused MFMA Toeplitz "
                         "FIR, 3 fast convolution (overlap-save kernel + batched FFT, two launches), "
                         "4 frequency-domain filter with frame-boundary correction (one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend for the barrier / max-over-ranks reduction (nccl = RCCL; "
                         "gloo lets several ranks share one GPU when rehearsing the N>1 path on a 1-GPU box)")
    args = ap.parse_args()

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

    def barrier():
        if world > 1:
            dist.barrier()

    eff_variant = 4 if args.variant == 0 else args.variant       # library's auto choice at 256 taps
    nframes = args.frames
    n = nframes * NFFT
    h = ya.fir_design_kaiser(TAPS, 0.2, 60.0)        # FirFilter::new_kaiser(256, 0.2, 60, 0)
    scale = 0.4
    x = torch.empty(n, dtype=torch.complex64, device=dev)
    y = torch.empty(n, dtype=torch.complex64, device=dev)
    stream = torch.cuda.current_stream()
    q = ya.FirFftStream(h, NFFT)
    q.set_scale(scale)
    q.set_variant(args.variant)
    q.set_stream(stream.cuda_stream)
    # each rank filters its own stream: rank r's samples are draws [r*2^40 + ...) of the generator
    ya.gen_complex_dev(SEED, n, out=x, first=rank << 40, stream=stream.cuda_stream)
    torch.cuda.synchronize()

    def step():
        q.execute_dev(x, nframes, y)

    # clock pre-conditioning (untimed, not part of the W warmup steps): keep the device busy with the same
    # kernel until the DVFS governor has ramped up from idle
    if args.prewarm_ms > 0:
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            for _ in range(10):
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
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                   # HIP events on the launch stream

    if world > 1:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64,
                         device=dev if args.dist_backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    # the same workload through the other algorithms, K steps each, reported beside the headline so all
    # three are on record (untimed w.r.t. `value`)
    def time_variant(v):
        qd = ya.FirFftStream(h, NFFT)
        qd.set_scale(scale)
        qd.set_variant(v)
        qd.set_stream(stream.cuda_stream)
        yd = torch.empty(n, dtype=torch.complex64, device=dev)
        for _ in range(max(args.warmup, 5)):
            qd.execute_dev(x, nframes, yd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            qd.execute_dev(x, nframes, yd)
        e1.record(stream)
        torch.cuda.synchronize()
        del yd
        return e0.elapsed_time(e1) / args.steps

    direct = fastconv = None
    if eff_variant == 4:
        d_ms = time_variant(2)            # fused MFMA Toeplitz FIR + FFT: the faster of the two direct forms
        direct = {"value": round(n / d_ms / 1e3, 3), "unit": "Msamples/s per GPU", "ms_per_step": round(d_ms, 4),
                  "kernel": "fir_crcf_mfma_kernel<68, true> (fused direct-form MFMA Toeplitz FIR + FFT, 16 B/sample)",
                  "fp32_tflops": round(FLOP_PER_SAMPLE * n / (d_ms / 1e3) / 1e12, 2)}
        c_ms = time_variant(3)
        fastconv = {"value": round(n / c_ms / 1e3, 3), "unit": "Msamples/s per GPU", "ms_per_step": round(c_ms, 4),
                    "kernel": "firfilt_fftconv_kernel<0> + fft4096_kernel<-1> (overlap-save FIR, then batched FFT; "
                              "32 B/sample)"}

    # variant 3 launches two kernels per step; the roofline object is about the dominant one (the overlap-save
    # FIR kernel), so time that kernel alone on the same input through FirFilter's kernel choice 4
    dom_ms = None
    if eff_variant == 3:
        qk = ya.FirFilter("crcf", h)
        qk.set_scale(scale)
        qk.set_kernel(4)
        qk.set_stream(stream.cuda_stream)
        yk = torch.empty(n, dtype=torch.complex64, device=dev)
        for _ in range(max(args.warmup, 5)):
            qk.execute_block_dev(x, n, yk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            qk.execute_block_dev(x, n, yk)
        e1.record(stream)
        torch.cuda.synchronize()
        dom_ms = e0.elapsed_time(e1) / args.steps
        del yk

    # in-run parity spot check (rank 0): last frame of the last step vs the oracle (f64 FIR + f64 FFT)
    # on the same input; its 255-sample halo is the preceding samples of the stream
    parity = None
    if rank == 0:
        try:
            from oracle import oracle
            f = nframes - 1
            halo = x[f * NFFT - (TAPS - 1): f * NFFT] if f > 0 else x[n - (TAPS - 1):]
            xs = torch.cat([halo, x[f * NFFT:(f + 1) * NFFT]]).cpu().numpy()
            yref = oracle.fir_block_f64("crcf", h, xs, scale=scale)[-NFFT:]
            truth = np.fft.fft(yref)
            got = y[f * NFFT:(f + 1) * NFFT].cpu().numpy()
            parity = float(np.linalg.norm(got - truth) / np.linalg.norm(truth))
        except Exception as e:            # the checker must never hide a bench result
            parity = f"unavailable: {e}"

    kernel_name = {1: "firfft_crcf_4096_slide_kernel", 2: "fir_crcf_mfma_kernel<68, true>",
                   3: "firfilt_fftconv_kernel<0> + fft4096_kernel<-1> (two launches per step)",
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
        samples = n * args.steps * world
        value = samples / elapsed / 1e6
        kern_s = dev_ms / 1e3 / args.steps           # average device time of one step (HIP events)
        if eff_variant == 3:
            # dominant kernel: 8*4096/3841 B read + 8 B written per sample
            dom_bytes = (8.0 * 4096 / 3841 + 8.0) * n
            dom_s = dom_ms / 1e3
            achieved = dom_bytes / dom_s / 1e9
        else:
            dom_bytes, dom_s = bytes_per_sample * n, kern_s
            achieved = dom_bytes / dom_s / 1e9
        out = {
            "metric": "Msamples/sec, 256-tap firfilt_crcf + 4096-pt FFT stream",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "firfilt_crcf 256-tap (kaiser 0.2/60dB, scale 0.4) -> 4096-pt forward FFT, "
                                   "streaming complex f32 (BASELINE configs[1] feeding configs[2])",
                       "algorithm": {3: "fast convolution (overlap-save, 4096-pt blocks) + batched FFT",
                                     4: "frequency-domain filter: FFT{h}.FFT{frame} + FFT{frame-boundary correction}"
                                     }.get(eff_variant, "fused direct-form FIR + FFT"),
                       "samples_per_step_per_gpu": n, "frames_per_step": nframes, "nfft": NFFT, "taps": TAPS,
                       "parallelism": f"{world} independent stream(s), no collective",
                       "kernel": kernel_name, "variant": args.variant, "prewarm_ms": args.prewarm_ms},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, bytes per launch; "
                                           "profiles/traffic.json" if traffic else None,
                         "kernel": kernel_name.split(" + ")[0],
                         "kernel_ms": round(dom_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": int(dom_bytes),
                         "step": {"ms": round(kern_s * 1e3, 4), "algorithmic_bytes": bytes_per_sample * n,
                                  "achieved_GBps": round(bytes_per_sample * n / kern_s / 1e9, 2),
                                  "frac_of_hbm_peak": round(bytes_per_sample * n / kern_s / 1e9 / HBM_PEAK_GBS, 4)},
                         "note": {3: "fast convolution: overlap-save FIR kernel + batched FFT, FIR output stream crosses HBM once",
                                  4: "one launch per step; the stream is read once and the spectra written once"
                                  }.get(eff_variant, "direct-form 256-tap crcf is FP32-ALU bound (64 flop/B); see fp32"),
                         "fp32": {"achieved_tflops": round(flop_per_sample * n / kern_s / 1e12, 2),
                                  "peak_tflops": FP32_PEAK_TFLOPS, "flop_per_sample": flop_per_sample,
                                  "frac": round(flop_per_sample * n / kern_s / 1e12 / FP32_PEAK_TFLOPS, 4)}},
            "parity_rel_l2_vs_f64": parity,
            "direct_form": direct,
            "fast_convolution": fastconv,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            out["cpu_baseline"] = cpu_baseline(oracle.fir_design_kaiser(TAPS, 0.2, 60.0), scale)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
