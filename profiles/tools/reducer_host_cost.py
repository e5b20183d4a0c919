"""Host cost of each part of a StatsReducer collective on a one-rank RCCL group (MI355X box), with the stream kept busy the way
a bench step keeps it (~125 us of kernels per iteration): fold, async all-reduce, and the completion query -- at a cadence of
one collective per step and one per 16 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from omniisaacgymenvs_amd.utils import distributed as od

n = 1 << 20
rew = torch.randn(n, device="cuda"); reset = (torch.rand(n, device="cuda") < 0.1).long(); prog = torch.randint(0, 1000, (n,), device="cuda")
x = torch.zeros(1 << 26, device="cuda")  # 256 MB: x.add_ is a ~90 us kernel


def run(every, iters=320):
    red = od.StatsReducer("cuda", every=every, force=True, consume="poll")
    for _ in range(2 * every):
        red.step(rew, reset, prog)
    torch.cuda.synchronize()
    T = dict(query=0.0, accumulate=0.0, fold=0.0, allreduce=0.0)
    calls = 0
    t_begin = time.perf_counter()
    for i in range(iters):
        x.add_(1.0)
        t0 = time.perf_counter(); red._consume(block=False); t1 = time.perf_counter()
        red._accumulate(rew, reset, prog, None); t2 = time.perf_counter()
        red._calls += 1
        T["query"] += t1 - t0; T["accumulate"] += t2 - t1
        if red._calls % every == 0:
            red._consume()
            out = red._send[red._slot]
            t3 = time.perf_counter(); red._fold(out); t4 = time.perf_counter()
            red._work = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True); red._slot ^= 1; t5 = time.perf_counter()
            T["fold"] += t4 - t3; T["allreduce"] += t5 - t4
            calls += 1
    host = time.perf_counter() - t_begin
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_begin
    print(f"every {every:2d}: per COLLECTIVE (host us): fold {T['fold'] / calls * 1e6:.1f}, all_reduce {T['allreduce'] / calls * 1e6:.1f}; "
          f"per STEP: query {T['query'] / iters * 1e6:.1f}, accumulate {T['accumulate'] / iters * 1e6:.1f}; "
          f"loop: host {host / iters * 1e6:.1f} us/iter, wall {wall / iters * 1e6:.1f} us/iter")


t0 = time.perf_counter()
for _ in range(320):
    x.add_(1.0)
torch.cuda.synchronize()
print(f"no reducer: wall {(time.perf_counter() - t0) / 320 * 1e6:.1f} us/iter")
run(1)
run(16)
dist.destroy_process_group()
