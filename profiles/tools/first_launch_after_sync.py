"""Host-side cost of the first kernel launch after torch.cuda.synchronize() against the following ones (us): a raw C-ABI launch
(oige_stream_copy on 4 KB) and a torch elementwise op.  Explains why a 20-step timed run reads ~7 % below a 200-step one."""
import time
import torch
from omniisaacgymenvs_amd import _lib

lib = _lib.load()
dev = "cuda:0"
a = torch.zeros(4096, dtype=torch.uint8, device=dev)
b = torch.zeros_like(a)
x = torch.zeros(1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    lib.oige_stream_copy(a.data_ptr(), b.data_ptr(), 4096, 1, st)
    x.add_(1.0)
for name, fn in (("oige_stream_copy", lambda: lib.oige_stream_copy(a.data_ptr(), b.data_ptr(), 4096, 1, st)), ("torch add_", lambda: x.add_(1.0))):
    for idle_ms in (0.0, 1.0, 20.0):
        rows = []
        for rep in range(5):
            torch.cuda.synchronize()
            if idle_ms:
                time.sleep(idle_ms * 1e-3)
            ts = []
            for i in range(4):
                t0 = time.perf_counter()
                fn()
                ts.append(round((time.perf_counter() - t0) * 1e6, 1))
            rows.append(ts)
        print(f"{name:18s} idle {idle_ms:5.1f} ms after sync: host us of launches 1..4:", rows)
