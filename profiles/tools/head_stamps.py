import ctypes, os, sys, json
import numpy as np
os.environ["OIGE_LIB"] = os.path.join(os.getcwd(), "omniisaacgymenvs_amd/variants/liboige_%s.so" % os.environ.get("VARIANT", "stamps"))
sys.argv = ["bench.py"]
import torch
sys.path.insert(0, os.getcwd()); import bench
from omniisaacgymenvs_amd import _lib
b = bench.ADAPTERS["AnymalTerrain"]("AnymalTerrain", 1 << 18, "cuda:0", 1234)
for _ in range(30):
    b.pre(); b.post()
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["OIGE_LIB"])
n = 4096 * 16
out = (ctypes.c_ulonglong * n)()
lib.oige_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.oige_debug_stamps(out, n) == 0
s = np.frombuffer(out, dtype=np.uint64).reshape(4096, 16)[:, :9].astype(np.int64)
print(os.environ.get("VARIANT", "stamps"), "wave lifetime mean", (s[:, 8] - s[:, 0]).mean(), "before D1", (s[:, 7] - s[:, 0]).mean(), "D1", (s[:, 8] - s[:, 7]).mean())
