#!/usr/bin/env python3
"""Pipelined vs plain block calls of the headline stream object (GPU box); run under
rocprofv3 --kernel-trace to see whether consecutive block kernels overlap.
usage: python tools/kb_pipe.py [pipelined=1] [steps=4]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, yagi_amd as ya
piped = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
ya.gen_complex_dev(7, NB * n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
q = ya.FirFftStream(h); q.set_scale(0.4); q.set_variant(4); q.set_stream(st.cuda_stream)
q.set_pipeline(bool(piped))
xp, yp = x.data_ptr(), y.data_ptr()
def run():
    for b in range(NB):
        q.execute_dev(xp + 8 * n * b, NF, yp + 8 * n * b)
    q.join()
for _ in range(30): run()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps): run()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue time: {(t1 - t0) / steps / NB * 1e6:.1f} us per execute_dev call (GPU idle at start)", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(steps): run()
e1.record(st); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f"pipelined={piped}: {ms:.4f} ms per 2^28 = {ms / 16 * 1000:.2f} us per 2^24", flush=True)
