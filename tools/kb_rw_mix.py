#!/usr/bin/env python3
"""What plain streaming kernels reach on this device for a given read : write mix (GPU box): torch element-wise
kernels over 2^28 complex64 of input, HIP events.  The practical ceiling beside which the path's kernels are read."""
import torch

dev = torch.device("cuda")
n = 1 << 27
x = torch.randn(2 * n, dtype=torch.float32, device=dev).view(torch.complex64)   # n complex = 1 GiB
y = torch.empty(2 * n, dtype=torch.complex64, device=dev)                       # 2 GiB


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [
    ("read 1 : write 1 (copy)", lambda: y[:n].copy_(x), 16 * n),
    ("read 1 : write 2 (broadcast copy)", lambda: y.view(2, n).copy_(x.unsqueeze(0).expand(2, n)), 24 * n),
    ("read 2 : write 1 (add)", lambda: torch.add(x, y[n:], out=y[:n]), 24 * n),
    ("read 8 : write 1 (strided pick)", lambda: y[: n // 8].copy_(x.view(n // 8, 8).sum(dim=1)), 9 * n),
    ("read only (sum)", lambda: x.view(torch.float32).sum(), 8 * n),
    ("write only (fill)", lambda: y.fill_(1.0), 16 * n),
]
for name, fn, nbytes in cases:
    ms = timed(fn)
    print(f"{name:38s} {ms:8.4f} ms  {nbytes / ms / 1e9:7.2f} TB/s  = {nbytes / ms / 1e9 / 8.0:.3f} of 8 TB/s")
