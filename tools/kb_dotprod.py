#!/usr/bin/env python3
"""device-resident inner products (dotprod::{rrrf,crcf,cccf}) over 2^24 elements: GB/s of operand reads"""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, yagi_amd as ya
from yagi_amd import lib
n = 1 << 24
dev = torch.device("cuda")
a = torch.empty(n, dtype=torch.complex64, device=dev); b = torch.empty(n, dtype=torch.complex64, device=dev)
yv = torch.empty(4, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(1, n, out=a, stream=st.cuda_stream); ya.gen_complex_dev(2, n, out=b, stream=st.cuda_stream)
for name, bytes_el in (("rrrf", 8), ("crcf", 12), ("cccf", 16)):
    fn = getattr(lib, f"yagi_hip_dotprod_{name}_dev")
    call = lambda: fn(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, C.c_void_p(yv.data_ptr()), st.cuda_stream)
    for _ in range(10): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50): call()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"dotprod_{name} n=2^24: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gelem/s  {bytes_el * n / ms / 1e6:8.1f} GB/s")
