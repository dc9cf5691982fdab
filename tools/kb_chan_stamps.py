#!/usr/bin/env python3
"""phase shares of one half tile (8 steps) of firpfbch2_col_kernel<8,8,false> (config C5) from s_memtime stamps of the
DIAGNOSTIC build.  usage: YAGI_HIP_LIB=variants/libyagi_stamps.so python tools/kb_chan_stamps.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya

n = 1 << 26
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(0x59414749 + 5, n, out=x, stream=st.cuda_stream)
c = ya.FirPfbCh2.new_kaiser(256, 4, 60.0)
c.set_stream(st.cuda_stream)
for _ in range(20):
    c.analyzer_execute_dev(x, n // 128, y)
torch.cuda.synchronize()
buf = np.zeros(2048 * 4 * 12, np.uint64)
assert ya.lib.yagi_hip_debug_chan_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
s = buf.reshape(2048, 4, 12).astype(np.int64)[:1024]
names = ["FIR (4 pairs x 16 FMAs) + LDS write + next loads issued", "barrier", "pass 1 (radix 16)", "barrier",
         "pass 2 (radix 16, twiddles)", "barrier", "LDS read + global stores issued", "barrier"]
d = np.diff(s[:, :, :9], axis=2)
tot = s[:, :, 8] - s[:, :, 0]
print(f"one half tile (8 steps: 1024 samples in, 2048 out): mean {tot.mean():.0f} cycles per workgroup")
for i, nm in enumerate(names):
    print(f"  {nm:58s} mean {d[:, :, i].mean():7.0f}  median {np.median(d[:, :, i]):7.0f}  {100 * d[:, :, i].mean() / tot.mean():5.1f} %")
