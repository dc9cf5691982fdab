#!/usr/bin/env python3
"""Occupancy probe: streamed 2^28 samples, us per 2^24-sample block, for the plain 4096-point transform and the
frequency-domain stream kernel at 4 / 3 / 2 workgroups per CU.
Needs a diagnostic build of the library (not in the tree) in which the two launches take their dynamic-LDS size from
the environment -- in fft_kernels.hip `fft4096_kernel<-1><<<grid, 256, getenv("YG_X") ? atoi(getenv("YG_X")) : 0, st>>>`
and the same third launch argument in freq_kernels.hip's `firfft_crcf_4096_freq_kernel<<<...>>>`; point YAGI_HIP_LIB
at it.  Output of the round-2 run: profiles/r02_occupancy_probe.txt."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, yagi_amd as ya
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(7, NB * n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
q = ya.FirFftStream(h); q.set_scale(0.4); q.set_variant(4); q.set_stream(st.cuda_stream)
plan = ya.Fft(4096, ya.Direction.Forward)
xp, yp = x.data_ptr(), y.data_ptr()
def run_q():
    for b in range(NB): q.execute_dev(xp + 8 * n * b, NF, yp + 8 * n * b)
def run_f():
    for b in range(NB): plan.run_batch_dev(xp + 8 * n * b, yp + 8 * n * b, NF, st.cuda_stream)
def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / NB * 1000
for rep in range(2):
    for xb, label in ((0, "4 WG/CU"), (13000, "3 WG/CU"), (24000, "2 WG/CU")):
        os.environ["YG_X"] = str(xb)
        print(f"{label}: fft4096 {timed(run_f):7.2f} us   freq kernel {timed(run_q):7.2f} us per 2^24", flush=True)
