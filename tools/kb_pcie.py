#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline stream through the host-pointer entry point (pageable numpy buffers)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import yagi_amd as ya
n = 1 << 24
rng = np.random.default_rng(0)
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
q = ya.FirFftStream(ya.fir_design_kaiser(256, 0.2, 60.0)); q.set_scale(0.4)
for _ in range(3): q.execute(x)
t0 = time.perf_counter()
for _ in range(10): q.execute(x)
dt = (time.perf_counter() - t0) / 10
print(f"host-pointer execute (H2D + kernel + D2H, pageable): {dt*1e3:.2f} ms per 2^24 samples = {n/dt/1e6:.0f} Msamples/s, {16*n/dt/1e9:.1f} GB/s over PCIe")
