#!/usr/bin/env python3
"""Print VGPR / SGPR / LDS / scratch of every kernel in csrc/*.hip (device-only compile, reads the .s metadata)."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "omniisaacgymenvs_amd", "csrc")
only = [a for a in sys.argv[1:] if not a.startswith("-D")]
defines = [a for a in sys.argv[1:] if a.startswith("-D")]  # e.g. -DOIGE_HAND_PRE_NT=512
with tempfile.TemporaryDirectory() as td:
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        base = os.path.basename(src)[:-4]
        if only and base not in only:
            continue
        s = os.path.join(td, base + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "--cuda-device-only", "-S", src, "-o", s] + defines, stderr=subprocess.DEVNULL)
        txt = open(s).read()
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", txt, flags=re.S):
            blk = m.group(0)
            g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
            name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
            print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
