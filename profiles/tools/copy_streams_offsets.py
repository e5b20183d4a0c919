"""Does the copy rate through many concurrent streams depend on how the streams' start addresses are aligned to each other?
The same 28-stream copy with sub-array sizes that are a multiple of 2 MiB, of 64 KiB + odd 4 KiB, of 256 B + odd ..."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from omniisaacgymenvs_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
streams = 28
base = 36 * 1024 * 1024
for extra in (0, 4096, 4096 * 17, 256, 256 * 3, 256 * 37, 2048 * 5, 65536 * 3, 1024 * 1024):
    per = base + extra
    n = per * streams
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"28 streams, sub-array = 36 MiB + {extra:8d} B: {ms*1e3:.1f} us  {2*n/ms/1e6:.0f} GB/s (read+write)", flush=True)
    del src, dst
