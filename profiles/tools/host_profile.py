"""cProfile of the host side of one task step (small N, so the GPU is never the limiter)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
task = sys.argv[1] if len(sys.argv) > 1 else "Ant"
b = bench.ADAPTERS[task](task, 4096, "cuda:0", 1234)
after = getattr(b, "after_post", lambda: None)
def loop(k):
    for _ in range(k):
        b.pre(); b.post(); after()
loop(50); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(500); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print(f"{task}: {dt*1e6:.1f} us per step (host-bound, 4096 envs)")
pr = cProfile.Profile(); pr.enable(); loop(300); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
