#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi) while a kernel loops: stream kernel on random / zero input, bare FFT, copy."""
import subprocess, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, yagi_amd as ya
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
q = ya.FirFftStream(h); q.set_scale(0.4); q.set_variant(4); q.set_stream(st.cuda_stream); q.set_pipeline(True)
plan = ya.Fft(4096, ya.Direction.Forward)
xp, yp = x.data_ptr(), y.data_ptr()
def run_stream():
    for b in range(NB):
        q.execute_dev(xp + 8 * n * b, NF, yp + 8 * n * b)
    q.join()
def run_fft():
    for b in range(NB):
        plan.run_batch_dev(xp + 8 * n * b, yp + 8 * n * b, NF, st.cuda_stream)
def run_copy():
    y.copy_(x)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("Power" in l or "sclk" in l or "mclk" in l or "fclk" in l)]
        return " | ".join(keep)
    except Exception as e:
        return f"rocm-smi unavailable: {e}"
def measure(name, fn, secs=4.0):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    stop = False
    res = []
    def sampler():
        time.sleep(1.5)
        res.append(smi())
        res.append(smi())
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); reps = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    while time.perf_counter() - t0 < secs:
        for _ in range(20): fn()
        reps += 20
        torch.cuda.synchronize()
    e1.record(st); torch.cuda.synchronize()
    th.join()
    print(f"{name}: {e0.elapsed_time(e1) / reps / NB * 1e3:.2f} us per 2^24 (incl. sync gaps)")
    for r in res: print("    ", r)
ya.gen_complex_dev(7, NB * n, out=x, stream=st.cuda_stream); torch.cuda.synchronize()
print("idle:", smi())
measure("stream kernel, random input", run_stream)
measure("bare fft4096, random input", run_fft)
measure("copy 2 GiB -> 2 GiB", run_copy)
x.zero_(); torch.cuda.synchronize()
measure("stream kernel, zero input", run_stream)
