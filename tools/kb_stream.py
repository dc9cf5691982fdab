import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch, yagi_amd as ya
from oracle import oracle
h = oracle.fir_design_kaiser(256, 0.2, 60.0)
n = 1 << 24
dx = ya.gen_complex_dev(1, n); dy = ya.DeviceArray(n, np.complex64)
for v in (4,):
    q = ya.FirFftStream(h); q.set_scale(0.4); q.set_variant(v)
    for _ in range(300): q.execute_dev(dx, n // 4096, dy)
    ya.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    for _ in range(200): q.execute_dev(dx, n // 4096, dy)
    ya.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print(f"variant {v}: {ms:.4f} ms/step  {n/ms/1e6:.1f} Gsamples/s  {16*n/ms/1e9:.0f} GB/s")
