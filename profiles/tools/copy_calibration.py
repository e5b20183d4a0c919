"""Practical HBM read+write stream rate of the box (calibration for the roofline fractions): liboige's streaming
copy over 1..12 lock-step sub-arrays, and torch's copy_, at the Ant kernel's footprint and larger."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omniisaacgymenvs_amd import _lib
lib = _lib.load()
dev = 'cuda:0'
for mb in (288, 1152):
    n = mb * 1024 * 1024
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    for streams in (1, 2, 6, 12):
        for _ in range(5):
            assert lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), n, streams, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"copy {mb} MiB streams={streams}: {ms*1e3:.1f} us  {2*n/ms/1e6:.0f} GB/s (read+write)")
        assert torch.equal(src, dst)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"torch copy_ {mb} MiB: {2*n/ms/1e6:.0f} GB/s")
