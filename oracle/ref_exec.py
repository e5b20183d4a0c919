"""ORACLE tooling (container only): run the reference's OWN function source under local torch.

The reference's task modules cannot be imported here (``ModuleNotFoundError: omni`` -- an
ordinary Python error, nothing was denied).  Its pure-torch functions
can still be executed verbatim: this module reads a reference ``.py`` file as text, pulls one
function (or class method) out of the AST, and compiles exactly that source in a namespace
holding ``torch / numpy / math`` and -- only where the function calls the un-vendored omni
helpers -- the restated helpers of ``oracle/torch_utils.py``.

Nothing here copies reference text into the repo: the source is read from
``/root/reference`` at run time, used to produce golden *data* (tests/golden/*.npz via
oracle/gen_golden.py) and to pin the restatement (tests/test_oracle_pinned.py).  When
``/root/reference`` is absent (the GPU box) ``available()`` is False and callers skip.
"""
from __future__ import annotations

import ast
import math
import os
import textwrap
from typing import Callable, Dict, Optional

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("OIGE_REFERENCE_ROOT", "/root/reference")
PKG = os.path.join(REFERENCE_ROOT, "omniisaacgymenvs")


def available() -> bool:
    return os.path.isdir(PKG)


def _read(relpath: str) -> str:
    with open(os.path.join(PKG, relpath), "r") as f:
        return f.read()


def _find(tree: ast.AST, name: str, cls: Optional[str]):
    body = tree.body
    if cls is not None:
        for node in body:
            if isinstance(node, ast.ClassDef) and node.name == cls:
                body = node.body
                break
        else:
            raise KeyError(cls)
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return node
    raise KeyError(name)


def load_function(
    relpath: str,
    name: str,
    cls: Optional[str] = None,
    helpers: bool = False,
    extra: Optional[Dict[str, object]] = None,
    keep_decorators: bool = True,
) -> Callable:
    """Compile ``name`` from reference file ``relpath`` (relative to omniisaacgymenvs/)."""
    src = _read(relpath)
    node = _find(ast.parse(src), name, cls)
    lines = src.splitlines()
    first = node.lineno
    if keep_decorators and node.decorator_list:
        first = min(d.lineno for d in node.decorator_list)
    else:
        first = node.lineno
    text = textwrap.dedent("\n".join(lines[first - 1 : node.end_lineno]))
    ns: Dict[str, object] = {"torch": torch, "np": np, "numpy": np, "math": math}
    if helpers:
        from . import torch_utils as tu

        for h in tu.HELPER_NAMES:
            ns[h] = getattr(tu, h)
    if extra:
        ns.update(extra)
    # torch.jit.script needs real source lines -> go through a temp module file
    if "torch.jit.script" in text:
        import importlib.util
        import tempfile

        header = "import math\nimport numpy as np\nimport torch\nfrom typing import Tuple\n"
        if helpers:
            header += "from oracle.torch_utils import *\nfrom oracle.torch_utils import " + ", ".join(
                __import__("oracle.torch_utils", fromlist=["HELPER_NAMES"]).HELPER_NAMES
            ) + "\n"
        # sibling jit functions the target calls (e.g. normalize_angle) come along by name
        deps = (extra or {}).get("__deps__", [])
        dep_text = ""
        for dep_rel, dep_name in deps:
            dsrc = _read(dep_rel)
            dnode = _find(ast.parse(dsrc), dep_name, None)
            dfirst = min([dnode.lineno] + [d.lineno for d in dnode.decorator_list])
            dep_text += textwrap.dedent("\n".join(dsrc.splitlines()[dfirst - 1 : dnode.end_lineno])) + "\n\n"
        tmpdir = tempfile.mkdtemp(prefix="oige_ref_")
        path = os.path.join(tmpdir, f"ref_{name}.py")
        with open(path, "w") as f:
            f.write(header + "\n" + dep_text + text + "\n")
        spec = importlib.util.spec_from_file_location(f"_oige_ref_{name}_{abs(hash(path))}", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return getattr(mod, name)
    exec(compile(text, f"<reference:{relpath}:{name}>", "exec"), ns)
    return ns[name]  # type: ignore[return-value]


class Stub:
    """A bare ``self`` for running reference *methods*: attributes are set by the caller."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
