#!/usr/bin/env python3
"""Is the headline kernel clock-/power-limited?  The same launches on random and on all-zero input (identical
instruction stream and traffic; only the switching activity differs), plus the bare batched 4096-point FFT."""
import sys
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
def t(fn, reps=20):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / NB * 1e3
for rnd in range(2):
    for name, fill in (("random", None), ("zeros", 0)):
        if fill is None:
            ya.gen_complex_dev(7, NB * n, out=x, stream=st.cuda_stream)
        else:
            x.zero_()
        torch.cuda.synchronize()
        print(f"{name:7s}: stream kernel (pipelined) {t(run_stream):6.2f} us / 2^24   bare fft4096 {t(run_fft):6.2f} us / 2^24", flush=True)
