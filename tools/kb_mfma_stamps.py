#!/usr/bin/env python3
"""phase shares of the MFMA direct-form FIR kernel from the s_memtime stamps of the DIAGNOSTIC build
(make EXTRA=-DYG_STAMPS BUILD=build_stamps OUT=../../variants/libyagi_stamps.so).
usage: YAGI_HIP_LIB=variants/libyagi_stamps.so python tools/kb_mfma_stamps.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya

n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(0x59414749 + 2, n, out=x, stream=st.cuda_stream)
q = ya.FirFilter("crcf", ya.fir_design_kaiser(256, 0.2, 60.0))
q.set_kernel(3)
q.set_stream(st.cuda_stream)
for _ in range(30):
    q.execute_block_dev(x, n, y)
torch.cuda.synchronize()
buf = np.zeros(4096 * 4 * 8, np.uint64)
assert ya.lib.yagi_hip_debug_mfma_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
s = buf.reshape(4096, 4, 8).astype(np.int64)
names = ["stage span (global -> LDS)", "barrier", "MFMA phase (2 tasks x 272 MFMAs)", "barrier", "accumulators -> LDS image + barrier",
         "LDS image -> global stores issued", "store drain"]
d = np.diff(s, axis=2)
tot = s[:, :, 7] - s[:, :, 0]
print(f"wave lifetime mean {tot.mean():.0f} cycles (median {np.median(tot):.0f}); MFMA pipe time per wave = 544 x 32 = 17408")
for i, nm in enumerate(names):
    print(f"  {nm:42s} mean {d[:, :, i].mean():8.0f}  median {np.median(d[:, :, i]):8.0f}  {100 * d[:, :, i].mean() / tot.mean():5.1f} %")
span = s[:, :, 7].max() - s[:, :, 0].min()
print(f"launch span {span} cycles; sum of wave lifetimes / span = {tot.sum() / span:.1f} waves resident (of 3072 slots at 3 WG/CU)")
print(f"MFMA pipe busy = 16384 waves x 17408 / (1024 SIMDs x span) = {16384 * 17408 / (1024 * span):.3f}")
