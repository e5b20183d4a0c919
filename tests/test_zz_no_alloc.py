"""GPU: oige_compact_i64_static, the entry point that allocates nothing (its workspace is a static device array inside
the library), against oige_compact_i64 with a caller workspace -- ids and count bit-exact, on two calls in a row, the
second with a smaller input (the static array keeps the first call's contents), and the size limit."""
import ctypes

import pytest
import torch

from omniisaacgymenvs_amd import _lib

pytestmark = pytest.mark.gpu


def _both(flags):
    n = flags.shape[0]
    dev = flags.device
    ids_a = torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev)
    ids_b = torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev)
    cnt_a = torch.full((1,), -1, dtype=torch.int64, device=dev)
    cnt_b = torch.full((1,), -1, dtype=torch.int64, device=dev)
    _lib.compact(flags, _lib.new_workspace(n, dev), ids_a, cnt_a)
    _lib.compact_static(flags, ids_b, cnt_b)
    torch.cuda.synchronize()
    return ids_a, cnt_a, ids_b, cnt_b


def test_static_compaction_matches_the_workspace_entry_point_on_two_calls():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    for n, p in ((1 << 20) + 77, 0.3), (4097, 0.9), (257, 0.5), (1, 1.0), ((1 << 18), 0.001):  # big first, then smaller ones
        flags = (torch.rand(n, generator=g) < p).long().to(dev) * 3  # any non-zero value is a flag
        ids_a, cnt_a, ids_b, cnt_b = _both(flags)
        want = flags.nonzero().squeeze(-1)
        assert int(cnt_a) == int(cnt_b) == want.numel(), n
        assert torch.equal(ids_b, ids_a), n  # same ids AND the same untouched tail
        assert torch.equal(ids_b[: want.numel()], want), n


def test_static_compaction_of_zero_flags_sets_the_count_to_zero():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    flags = torch.ones(4, dtype=torch.int64, device=dev)  # valid pointers, n = 0: nothing is read
    ids = torch.full((4,), -7, dtype=torch.int64, device=dev)
    for fn, extra in ((lib.oige_compact_i64, (_lib.new_workspace(4, dev).data_ptr(),)), (lib.oige_compact_i64_static, ())):
        cnt = torch.full((1,), -1, dtype=torch.int64, device=dev)
        _lib.check(fn(flags.data_ptr(), 0, *extra, ids.data_ptr(), cnt.data_ptr(), _lib.current_stream(dev)), "compact n=0")
        torch.cuda.synchronize()
        assert int(cnt) == 0 and ids.tolist() == [-7] * 4


def test_static_compaction_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    cap = int(lib.oige_compact_static_max_envs())
    assert cap == 1 << 22
    dev = torch.device("cuda:0")
    flags = torch.zeros(16, dtype=torch.int64, device=dev)
    ids = torch.zeros(16, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = lib.oige_compact_i64_static(flags.data_ptr(), cap + 1, ids.data_ptr(), cnt.data_ptr(), None)
    assert rc == -1 and b"exceed the static scratch" in lib.oige_last_error()
    # the limit itself is accepted
    big = torch.zeros(cap, dtype=torch.int64, device=dev)
    big[-1] = 1
    big[12345] = 1
    ids = torch.zeros(cap, dtype=torch.int64, device=dev)
    _lib.compact_static(big, ids, cnt)
    torch.cuda.synchronize()
    assert int(cnt) == 2 and ids[:2].tolist() == [12345, cap - 1]


def test_static_compaction_is_capturable_after_one_warm_call():
    """the use case the header names is code under graph capture.  The first call on a device looks the scratch
    address up (hipGetSymbolAddress) and so happens outside the window; the call inside it is captured and replayed on new
    flags."""
    dev = torch.device("cuda:0")
    n = 5000
    g = torch.Generator(device="cpu").manual_seed(9)
    flags = (torch.rand(n, generator=g) < 0.2).long().to(dev)
    ids = torch.full((n,), -7, dtype=torch.int64, device=dev)
    cnt = torch.full((1,), -1, dtype=torch.int64, device=dev)
    _lib.compact_static(flags, ids, cnt)  # warm call: per-device look-up, code objects
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        _lib.compact_static(flags, ids, cnt)
    for p in (0.7, 0.0, 0.05):
        flags.copy_((torch.rand(n, generator=g) < p).long())
        ids.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        want = flags.nonzero().squeeze(-1)
        assert int(cnt) == want.numel()
        assert torch.equal(ids[: want.numel()], want) and bool((ids[want.numel():] == -7).all())
