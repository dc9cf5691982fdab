#!/usr/bin/env python3
"""FFT size sweep (GPU box): sustained time of Fft::run over 2^KB_TOTAL_LOG2 points (default 24 = 128 MiB in + 128 MiB out:
Infinity-Cache resident; 28 = 2 GiB + 2 GiB: streamed from and to HBM) for each size."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import yagi_amd as ya

n = 1 << int(os.environ.get("KB_TOTAL_LOG2", "24"))      # points per run
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(7, n, out=x, stream=st.cuda_stream)
sizes = [int(a) for a in sys.argv[1:]] or [16, 64, 100, 128, 256, 480, 512, 1024, 2048, 4096, 8192]
for N in sizes:
    plan = ya.Fft(N, ya.Direction.Forward)
    nb = n // N
    reps = 50 if n <= (1 << 24) else 12
    for _ in range(reps // 2):
        plan.run_batch_dev(x, y, nb, st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        plan.run_batch_dev(x, y, nb, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"fft N={N:5d} x {nb:8d}: {ms:8.4f} ms  {nb * N / ms / 1e6:8.1f} Gpoint/s  {16 * nb * N / ms / 1e6:8.1f} GB/s")
