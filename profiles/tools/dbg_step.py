import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench
from omniisaacgymenvs_amd.utils import distributed as od
b = bench.ADAPTERS["Ant"]("Ant", 1<<20, "cuda:0", 1234)
for _ in range(20): b.pre(); b.post()
torch.cuda.synchronize()
for trial in range(3):
    t0=time.perf_counter()
    for _ in range(200): b.pre(); b.post()
    torch.cuda.synchronize(); t1=time.perf_counter()
    t=b.task
    s=od.all_reduce_stats(od.local_stats(t.rew_buf, t.reset_buf, t.progress_buf, None))
    torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"trial {trial}: loop {1e3*(t1-t0)/200:.4f} ms/step, stats {1e3*(t2-t1):.2f} ms")
evs=[]
t0=time.perf_counter()
for _ in range(200):
    b.pre(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record(); b.post(); e1.record(); evs.append((e0,e1))
torch.cuda.synchronize(); t1=time.perf_counter()
print(f"with events: {1e3*(t1-t0)/200:.4f} ms/step")
pool=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for e0,e1 in pool[:5]: e0.record(); e1.record()
torch.cuda.synchronize()
t0=time.perf_counter()
for e0,e1 in pool:
    b.pre(); e0.record(); b.post(); e1.record()
torch.cuda.synchronize(); t1=time.perf_counter()
print(f"with pre-created events: {1e3*(t1-t0)/200:.4f} ms/step; post {sum(a.elapsed_time(c) for a,c in pool)/200:.4f} ms")
t0=time.perf_counter()
for i,(e0,e1) in enumerate(pool):
    b.pre()
    if i % 8 == 0: e0.record()
    b.post()
    if i % 8 == 0: e1.record()
torch.cuda.synchronize(); t1=time.perf_counter()
print(f"events every 8th step: {1e3*(t1-t0)/200:.4f} ms/step; post {sum(a.elapsed_time(c) for a,c in pool[::8])/25:.4f} ms")
