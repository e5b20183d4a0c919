import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["OIGE_BENCH_FORCE_GROUP"] = "1"
import torch
from omniisaacgymenvs_amd.utils import distributed as od
orig_step = od.StatsReducer.step
log = []
def step(self, *a, **k):
    t0 = time.perf_counter()
    w = self._work is not None
    c0 = self.collectives
    orig_step(self, *a, **k)
    log.append((round((time.perf_counter() - t0) * 1e6, 1), w, self.collectives - c0))
od.StatsReducer.step = step
orig_ar = od.dist.all_reduce
def ar(*a, **k):
    t0 = time.perf_counter(); r = orig_ar(*a, **k); print("all_reduce us", round((time.perf_counter() - t0) * 1e6, 1), file=sys.stderr); return r
od.dist.all_reduce = ar
orig_fold = od.StatsReducer._fold
def fold(self, *a, **k):
    t0 = time.perf_counter(); r = orig_fold(self, *a, **k); print("fold us", round((time.perf_counter() - t0) * 1e6, 1), file=sys.stderr); return r
od.StatsReducer._fold = fold
sys.argv = ["bench.py", "--steps", "48", "--warmup", "5", "--no-cpu-baseline", "--no-graph-leg"]
import runpy
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
finally:
    print("step() us, pending-before, issued:", log[-48:], file=sys.stderr)
