import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, yagi_amd as ya
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(7, NB * n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
q = ya.FirFftStream(h); q.set_scale(0.4); q.set_variant(4); q.set_stream(st.cuda_stream)
xp, yp = x.data_ptr(), y.data_ptr()
for per in (1, 2, 4, 8, 16):   # blocks per launch
    nl = NB // per
    def run():
        for b in range(nl):
            q.execute_dev(xp + 8 * n * per * b, NF * per, yp + 8 * n * per * b)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{per:2d} blocks per launch: {ms:.4f} ms per 2^28 = {ms / 16 * 1000:.2f} us per 2^24", flush=True)
