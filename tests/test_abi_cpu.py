"""CPU: liboige_hip.so loads without a GPU, exports every symbol include/oige_hip.h declares, and
the ctypes structs of omniisaacgymenvs_amd/_lib.py have the layout the C compiler gives the header."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from omniisaacgymenvs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "oige_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oige_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/oige_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert lib.oige_version() == 1
    assert lib.oige_num_tiles(65) == 2 and lib.oige_workspace_ints(257) == 8 + 2 + 1
    assert lib.oige_tile_counts_u8_bytes(257) == 16 and lib.oige_tile_counts_u8_bytes(256 * 5) == 32 and lib.oige_tile_counts_u8_bytes(0) == 0


def test_struct_layouts_match_the_c_header(tmp_path):
    structs = {k: v for k, v in vars(_lib).items() if isinstance(v, type) and issubclass(v, ctypes.Structure)
               and v is not ctypes.Structure}
    cname = lambda n: "oige_" + re.sub(r"(?<!^)(?=[A-Z])", "_", n).lower().replace("reset_i_o", "reset_io")
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "oige_hip.h"\nint main(void){\n'
    for n in sorted(structs):
        src += f'  printf("{n} %zu\\n", sizeof({cname(n)}));\n'
        for fname, *_ in structs[n]._fields_:   # every field sits where the C compiler puts it (same names on both sides)
            src += f'  printf("{n}.{fname} %zu\\n", offsetof({cname(n)}, {fname}));\n'
    src += '  printf("lower %zu\\n", offsetof(oige_locomotion_params, dof_limits_lower));\n'
    src += '  printf("count %zu\\n", offsetof(oige_locomotion_reset_io, count));\n  return 0;\n}\n'
    c = tmp_path / "sz.c"
    c.write_text(src)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, cls in structs.items():
        assert int(out[n]) == ctypes.sizeof(cls), f"{n}: C {out[n]} vs ctypes {ctypes.sizeof(cls)}"
        for fname, *_ in cls._fields_:
            assert int(out[f"{n}.{fname}"]) == getattr(cls, fname).offset, f"{n}.{fname}: C offset {out[f'{n}.{fname}']} vs ctypes {getattr(cls, fname).offset}"
    assert int(out["lower"]) == _lib.LocomotionParams.dof_limits_lower.offset
    assert int(out["count"]) == _lib.LocomotionResetIO.count.offset


def test_bad_arguments_fail_loudly_without_a_gpu():
    lib = _lib.load()
    p = _lib.LocomotionParams()
    p.num_dof, p.num_sensors = 5, 1  # not a built configuration
    rc = lib.oige_locomotion_post_physics(ctypes.byref(p), None, None, 10, _lib.STAGE_ALL, None)
    assert rc == -1 and b"unsupported" in lib.oige_last_error()
    with pytest.raises(_lib.OigeError):
        _lib.check(rc, "oige_locomotion_post_physics")
    import torch
    with pytest.raises(_lib.OigeError, match="no CPU fallback"):
        _lib.require_device_tensor(torch.zeros(4), torch.float32, (4,), "obs_buf")


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under omniisaacgymenvs_amd/ may import it."""
    pkg = os.path.join(ROOT, "omniisaacgymenvs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)


def test_every_entry_point_rejects_null_arguments_without_touching_the_gpu():
    """All-NULL / zero arguments: every compute entry point must come back with a negative code and a message (never a
    crash, never a launch).  Validation runs before any HIP call, so this is safe without a device."""
    lib = _lib.load()
    pure = {"oige_version", "oige_num_tiles", "oige_workspace_ints", "oige_tile_counts_u8_bytes", "oige_last_error", "oige_compact_static_max_envs",
            "oige_episode_stats_blocks", "oige_episode_stats_fold_scratch"}
    checked = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        if name in pure:
            continue
        argv = [None if (a is _lib._P or (isinstance(a, type) and issubclass(a, ctypes._Pointer))) else 0 for a in args]
        rc = getattr(lib, name)(*argv)
        assert rc < 0, f"{name} accepted NULL arguments (rc={rc})"
        msg = lib.oige_last_error()
        assert msg and len(msg) > 5, name
        checked += 1
    assert checked >= 20


def test_stale_check_covers_every_source_and_header():
    """build.is_stale() must see every file the library is built from (oige_noise.hpp / oige_scan.hpp were
    missing): every #include "..." under csrc/ resolves to a file in build.dependencies(), and every .hip is compiled."""
    import glob
    import re

    from omniisaacgymenvs_amd import build as b

    deps = {os.path.realpath(d) for d in b.dependencies()}
    hips = sorted(glob.glob(os.path.join(b.CSRC, "*.hip")))
    assert sorted(os.path.basename(h) for h in hips) == sorted(b.SOURCES)
    for src in hips + glob.glob(os.path.join(b.CSRC, "*.hpp")):
        assert os.path.realpath(src) in deps
        for inc in re.findall(r'#include\s+"([^"]+)"', open(src).read()):
            cands = [os.path.join(b.CSRC, inc), os.path.join(b.HERE, "..", "include", inc), os.path.join(b.CSRC, os.path.dirname(inc), os.path.basename(inc))]
            hit = [c for c in cands if os.path.exists(c)]
            assert hit, (src, inc)
            assert os.path.realpath(hit[0]) in deps, (src, inc)
