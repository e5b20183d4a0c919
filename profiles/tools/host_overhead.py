"""Where the host time of a step goes at the reference's own size (4096 envs, launch-bound): cProfile over N steps of
pre_physics_step + post_physics_step through the RLTask mirror.  Usage: python profiles/tools/host_overhead.py [Task] [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Ant"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    b = bench.ADAPTERS[name](name, 4096, "cuda:0", 1234)
    for _ in range(50):
        b.pre(), b.post()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.pre(), b.post()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e6
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        b.pre(), b.post()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print(f"{name}: {wall:.1f} us per step (wall, unprofiled)")
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:30]))


if __name__ == "__main__":
    main()
