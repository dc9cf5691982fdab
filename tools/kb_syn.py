import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, yagi_amd as ya
n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev); y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(9, n, out=x, stream=st.cuda_stream)
for M, m in [(16, 4), (64, 8), (256, 4), (1024, 2), (100, 4)]:
    c = ya.FirPfbCh.new_kaiser(M, m, 60.0); c.set_stream(st.cuda_stream)
    fn = lambda: c.synthesizer_execute_dev(x, n // M, y)
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): fn()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"firpfbch synthesizer M={M:5d} m={m}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {16 * n / ms / 1e6:8.1f} GB/s")
