"""Host cost of each part of StatsReducer.step on a one-rank RCCL group (MI355X box): wait, fold, snapshot copy, async all-reduce."""
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
red = od.StatsReducer("cuda", force=True)
x = torch.zeros(1 << 24, device="cuda")
for _ in range(20):
    red.step(rew, reset, prog)
torch.cuda.synchronize()
T = dict(consume=0.0, fold=0.0, copy=0.0, allreduce=0.0, total=0.0)
N = 300
for _ in range(N):
    x.add_(1.0)  # some GPU work between the steps (~20 us), so the stream is never empty
    t0 = time.perf_counter(); red._consume(); t1 = time.perf_counter()
    red._fold(rew, reset, prog, None); t2 = time.perf_counter()
    out = red._send[red._slot]; out.copy_(red.acc, non_blocking=True); t3 = time.perf_counter()
    red._work = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True); red._slot ^= 1; t4 = time.perf_counter()
    T["consume"] += t1 - t0; T["fold"] += t2 - t1; T["copy"] += t3 - t2; T["allreduce"] += t4 - t3; T["total"] += t4 - t0
torch.cuda.synchronize()
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "us per step (host)")
t0 = time.perf_counter()
for _ in range(N):
    red.step(rew, reset, prog)
torch.cuda.synchronize()
print("back-to-back step():", round((time.perf_counter() - t0) / N * 1e6, 1), "us per call incl. GPU")
dist.destroy_process_group()
