import sys, os, torch
sys.path.insert(0, os.getcwd())
from omniisaacgymenvs_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
for mb in (1008,):
    n = mb * 1024 * 1024
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for streams in (1, 4, 8, 12, 16, 24, 32, 48, 64):
        if n % (16*streams): continue
        for _ in range(3):
            assert lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"copy {mb} MiB streams={streams}: {ms*1e3:.1f} us  {2*n/ms/1e6:.0f} GB/s (read+write)", flush=True)
