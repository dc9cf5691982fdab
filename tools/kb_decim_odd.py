#!/usr/bin/env python3
"""decimators with a decimation factor that is not a power of two, 2^24 complex inputs (GPU box)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import yagi_amd as ya

n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(3, n, out=x, stream=st.cuda_stream)
for M, L in ((3, 49), (3, 97), (5, 161), (6, 193), (7, 225), (12, 385)):
    q = ya.FirDecimationFilter("crcf", M, ya.fir_design_kaiser(L, 0.4 / M, 60.0))
    q.set_stream(st.cuda_stream)
    ny = n // M
    fn = lambda: q.execute_block_dev(x, ny, y)
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"firdecim_crcf M={M:2d} L={L:4d}: {ms:8.4f} ms  {ny * M / ms / 1e6:8.1f} Gsamples/s in")
