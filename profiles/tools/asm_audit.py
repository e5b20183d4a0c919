#!/usr/bin/env python3
"""Static audit of the gfx950 assembly of every kernel in csrc/*.hip for the two memory-ordering traps found in round 2:

  * loads issued BEHIND stores: loads and stores share one in-order counter (vmcnt), so the wait for such a load is also a
    wait for every earlier store's round trip to HBM;
  * full waits (`s_waitcnt vmcnt(0)`) behind stores: the join of a conditional block that holds a vector load, or a
    run-time-indexed read of a kernel-argument array (a VECTOR load from the argument block), drains everything in flight.

Linear scan of the instruction stream (control flow ignored: a count, not a proof).  Usage:
    python profiles/tools/asm_audit.py [file stems...] [-DNAME=VALUE ...]
"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "omniisaacgymenvs_amd", "csrc")
seq = None  # --seq=<substring of a kernel name>: print that kernel's memory-operation sequence instead of the table
for a in sys.argv[1:]:
    if a.startswith("--seq="):
        seq = a[6:]
only = [a for a in sys.argv[1:] if not a.startswith("-")]
defines = [a for a in sys.argv[1:] if a.startswith("-D")]

if seq is None:
    print(f"{'kernel':72s} {'loads':>6s} {'stores':>6s} {'ld>st':>6s} {'wait0>st':>8s}")
with tempfile.TemporaryDirectory() as td:
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        base = os.path.basename(src)[:-4]
        if only and base not in only:
            continue
        s = os.path.join(td, base + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"), src, "-o", s] + defines,
                              stderr=subprocess.DEVNULL)
        name, seen_store, loads, stores, ld_after, wait_after = None, False, 0, 0, 0, 0
        ops = []
        for line in open(s):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name, seen_store, loads, stores, ld_after, wait_after = m.group(1), False, 0, 0, 0, 0
                ops = []
                continue
            if name is None:
                continue
            t = line.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                loads += 1
                ld_after += seen_store
                ops.append("L")
            elif t.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
                stores += 1
                seen_store = True
                ops.append("S")
            elif t.startswith("s_waitcnt") and "vmcnt(" in t:
                wait_after += seen_store and "vmcnt(0)" in t
                ops.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("s_barrier"):
                ops.append("|")
            elif "Loop Header" in t:
                ops.append("{loop")
            elif t.startswith(".Lfunc_end"):
                dn = subprocess.check_output(["c++filt", name], text=True).strip().replace("oige::", "")
                if seq is None:
                    print(f"{dn[:72]:72s} {loads:6d} {stores:6d} {ld_after:6d} {wait_after:8d}")
                elif seq in dn:
                    print(dn + ":\n  " + " ".join(ops))
                name = None
