"""Host-side time of the first pre_physics_step after the bench's untimed statistics reduction, with and without a short busy
spin in between (the reduction's blocking read-backs leave the core in a low-power state and Python's working set cold)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench as B
from omniisaacgymenvs_amd.utils import distributed as od

def spin(ms):
    s = time.perf_counter()
    while time.perf_counter() - s < ms * 1e-3:
        pass

for label, between in (("nothing", lambda: None), ("spin 0.5 ms", lambda: spin(0.5)), ("spin 3 ms", lambda: spin(3.0)),
                       ("one untimed extra step + sync", None)):
    b = B.LocomotionBench("Ant", 1 << 18, "cuda:0", 1234)
    t = b.task
    for _ in range(5):
        b.pre(); b.post()
    torch.cuda.synchronize()
    od.all_reduce_stats(od.local_stats(t.rew_buf, t.reset_buf, t.progress_buf, None)); torch.cuda.synchronize()
    if between is None:
        b.pre(); b.post(); torch.cuda.synchronize()
    else:
        between()
    t0 = time.perf_counter(); b.pre(); t1 = time.perf_counter(); b.post(); t2 = time.perf_counter(); b.pre(); t3 = time.perf_counter()
    print(f"{label:32s} pre {1e6*(t1-t0):7.1f}  post {1e6*(t2-t1):7.1f}  2nd pre {1e6*(t3-t2):7.1f}")
    del b, t
