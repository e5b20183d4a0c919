#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small CSV/JSON summaries (runs on the GPU box after the
profiler; the raw traces are too big to carry back).

  kernel stats : <dir>/**/*_kernel_stats.csv       -> <out>/<task>_kernel_stats.csv   (names cut to 100 chars, top 12 rows)
  counters     : <dir>/**/*_counter_collection.csv -> per-kernel mean of each counter per dispatch (JSON)
HBM bytes per launch follow MI355X_MICROARCH.md "HBM": rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
FETCH_SIZE counts 64 B per 128 B request for wide streaming reads, so it is doubled; WRITE_SIZE is taken as is.
(Checked here on the Ant kernel: 2 x FETCH = 277 MB vs 276.8 MB of algorithmic reads, WRITE = 306.8 MB vs 306.2 MB.)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


DEFAULT_ENVS = {"Ant": 1 << 20, "Humanoid": 1 << 19, "ShadowHand": 1 << 20, "AnymalTerrain": 1 << 18, "Anymal": 1 << 20,
                "AllegroHand": 1 << 20, "Quadcopter": 1 << 20, "BallBalance": 1 << 20, "Ingenuity": 1 << 20,
                "Crazyflie": 1 << 20, "FrankaCabinet": 1 << 20, "TomoDJCabinet": 1 << 20, "TomoSPCCabinet": 1 << 20,
                "TomoDJnewhandCabinet": 1 << 19, "TomoDJnewhandTendonCabinet": 1 << 19, "TomoSPCnewhandCabinet": 1 << 19}  # bench.py defaults


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name[:100]


def kernel_stats(src_dir: str, dst: str) -> None:
    files = glob.glob(os.path.join(src_dir, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counters(src_dir: str) -> dict:
    out = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(src_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in out.items()}


def main() -> None:
    root, tag = sys.argv[1], sys.argv[2]
    dst_dir = os.path.join("gpurun_out", f"summary_{tag}")
    os.makedirs(dst_dir, exist_ok=True)
    pmc = {}
    for task in DEFAULT_ENVS:
        if not os.path.isdir(os.path.join(root, task)):
            continue
        kernel_stats(os.path.join(root, task), os.path.join(dst_dir, f"{task}_kernel_stats.csv"))
        merged = defaultdict(dict)
        for kind in ("fetch", "write"):
            for k, cs in counters(os.path.join(root, f"{task}_{kind}")).items():
                if "oige::" in k:
                    merged[k].update(cs)
        pmc[task] = merged
        for ext in ("bench.log", "fetch.log", "write.log"):
            p = os.path.join(root, f"{task}.{ext}")
            if os.path.exists(p):
                txt = open(p).read()[-3000:]
                open(os.path.join(dst_dir, f"{task}.{ext}"), "w").write(txt)
    json.dump(pmc, open(os.path.join(dst_dir, "pmc_counters.json"), "w"), indent=1)
    # HBM bytes per launch of the dominant kernel: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes
    # (AnymalTerrain's post_physics_step is several kernels: their bytes are summed, as bench.py times them together)
    dominant = {"Ant": ["locomotion_post_kernel"], "Humanoid": ["locomotion_post_kernel"], "ShadowHand": ["hand_post_kernel"],
                "AnymalTerrain": ["anymal_head_kernel", "anymal_scan_kernel", "anymal_flags_kernel", "anymal_extras_kernel",
                                  "anymal_scan_norm_kernel"],
                "Anymal": ["anymal_flat_post_kernel"], "AllegroHand": ["allegro_post_kernel"],
                "Quadcopter": ["quadcopter_post_kernel"], "BallBalance": ["ball_balance_post_kernel"],
                "Ingenuity": ["ingenuity_post_kernel"], "Crazyflie": ["crazyflie_post_kernel"],
                "FrankaCabinet": ["franka_cabinet_post_kernel"],
                "TomoDJCabinet": ["franka_cabinet_post_kernel"], "TomoSPCCabinet": ["franka_cabinet_post_kernel"],
                "TomoDJnewhandCabinet": ["newhand_cabinet_post_kernel"], "TomoDJnewhandTendonCabinet": ["newhand_cabinet_post_kernel"],
                "TomoSPCnewhandCabinet": ["newhand_cabinet_post_kernel"]}
    traffic = {}
    for task, ks in pmc.items():
        fetch = write = 0.0
        names = []
        for k, cs in ks.items():
            if any(d in k for d in dominant[task]) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fetch += cs["FETCH_SIZE"]["mean"] * 1024.0
                write += cs["WRITE_SIZE"]["mean"] * 1024.0
                names.append(k)
        if names:
            traffic[task] = {"kernel": " + ".join(names), "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                             "hbm_bytes_per_launch": 2.0 * fetch + write, "envs": DEFAULT_ENVS[task]}
    # the same for the pre-physics kernels (one entry "<Task>:pre" each): every oige kernel of the task's run that is not one of
    # its post kernels -- the *_pre_kernel, AnymalTerrain's torques kernel, the scan / finalize / extras launches that ride along
    for task, ks in pmc.items():
        fetch = write = 0.0
        names = []
        for k, cs in ks.items():
            is_post = any(d in k for d in dominant[task])
            if not is_post and ("_pre_kernel" in k or "torques" in k) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fetch += cs["FETCH_SIZE"]["mean"] * 1024.0
                write += cs["WRITE_SIZE"]["mean"] * 1024.0
                names.append(k)
        if names:
            traffic[task + ":pre"] = {"kernel": " + ".join(names), "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                                      "hbm_bytes_per_launch": 2.0 * fetch + write, "envs": DEFAULT_ENVS[task]}
    json.dump(traffic, open(os.path.join(dst_dir, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
