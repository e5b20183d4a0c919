"""Reads the reference's configuration files as they are: ``cfg/config.yaml`` + ``cfg/task/<Task>.yaml``
(+ ``cfg/train/<Task>PPO.yaml`` when present) composed and resolved into the plain dict that
``omegaconf_to_dict(cfg)`` hands to ``initialize_task`` (scripts/rlgames_train.py:107-133, utils/hydra_cfg/reformat.py:35-43).

hydra and omegaconf are not installed here, and the path needs very little of them: the ``defaults`` lists of
cfg/config.yaml:78-82 and of task files that inherit (cfg/task/ShadowHandOpenAI_LSTM.yaml:3-5), ``key=value`` overrides,
``${...}`` interpolations (absolute and dot-relative), and the four resolvers the reference registers
(utils/hydra_cfg/hydra_utils.py:35-44): ``eq``, ``contains``, ``if``, ``resolve_default``.  Anything else in a file raises.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import yaml

Path = Tuple[Union[str, int], ...]


class ConfigError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------- resolvers
def _lower(x: Any) -> str:
    return str(x).lower()


RESOLVERS = {
    # hydra_utils.py:36-44, same argument order
    "eq": lambda x, y: _lower(x) == _lower(y),
    "contains": lambda x, y: _lower(x) in _lower(y),
    "if": lambda pred, a, b: a if pred else b,
    "resolve_default": lambda default, arg: default if arg == "" else arg,
}


# ------------------------------------------------------------------------------------------------------- composition
def _load_yaml(path: str) -> Dict[str, Any]:
    with open(path) as f:
        data = yaml.safe_load(f)
    if data is None:
        return {}
    if not isinstance(data, dict):
        raise ConfigError(f"{path}: top level must be a mapping")
    return data


def _merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    """Deep merge, ``over`` wins; lists and scalars are replaced (omegaconf's merge)."""
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _load_group_file(cfg_dir: str, group: str, name: str, _seen: Tuple[str, ...] = ()) -> Dict[str, Any]:
    """One file of a config group, with the same-group inheritance of its own ``defaults`` list."""
    path = os.path.join(cfg_dir, group, name + ".yaml")
    if path in _seen:
        raise ConfigError(f"{path}: circular defaults")
    if not os.path.isfile(path):
        raise ConfigError(f"no {group} config named {name!r} ({path})")
    body = _load_yaml(path)
    defaults = body.pop("defaults", None)
    if defaults is None:
        return body
    out: Dict[str, Any] = {}
    saw_self = False
    for d in defaults:
        if d == "_self_":
            out, saw_self = _merge(out, body), True
        elif isinstance(d, str):
            out = _merge(out, _load_group_file(cfg_dir, group, d, _seen + (path,)))
        else:
            raise ConfigError(f"{path}: unsupported defaults entry {d!r}")
    return out if saw_self else _merge(out, body)


def _parse_overrides(overrides: Union[None, Dict[str, Any], Iterable[str]]) -> Dict[str, Any]:
    if overrides is None:
        return {}
    if isinstance(overrides, dict):
        return dict(overrides)
    out = {}
    for item in overrides:  # hydra's command line: key=value, value read as a YAML scalar
        if "=" not in item:
            raise ConfigError(f"override {item!r}: expected key=value")
        k, v = item.split("=", 1)
        out[k.lstrip("+")] = yaml.safe_load(v) if v != "" else ""
    return out


def _set_dotted(cfg: Dict[str, Any], key: str, value: Any) -> None:
    node = cfg
    parts = key.split(".")
    for p in parts[:-1]:
        nxt = node.get(p)
        if not isinstance(nxt, dict):
            nxt = node[p] = {}
        node = nxt
    node[parts[-1]] = value


# ------------------------------------------------------------------------------------------------------ interpolation
def _matching_brace(s: str, start: int) -> int:
    """Index of the ``}`` closing the ``${`` whose ``{`` sits at ``start``."""
    depth, i, quote = 0, start, None
    while i < len(s):
        c = s[i]
        if quote:
            if c == quote:
                quote = None
        elif c in "\"'":
            quote = c
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ConfigError(f"unbalanced interpolation in {s!r}")


def _split_args(s: str) -> List[str]:
    args, depth, quote, cur = [], 0, None, []
    for c in s:
        if quote:
            cur.append(c)
            if c == quote:
                quote = None
            continue
        if c in "\"'":
            quote = c
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
        elif c == "," and depth == 0:
            args.append("".join(cur))
            cur = []
            continue
        cur.append(c)
    args.append("".join(cur))
    return args


class _Resolver:
    def __init__(self, root: Dict[str, Any]):
        self.root = root
        self._active: List[Path] = []
        self._active_text: Dict[Path, str] = {}

    # -- node access
    def _get(self, path: Path, asked_from: Path) -> Any:
        node: Any = self.root
        for i, p in enumerate(path):
            if isinstance(node, dict) and p in node:
                node = node[p]
            elif isinstance(node, list) and isinstance(p, int) and p < len(node):
                node = node[p]
            else:
                raise ConfigError(f"{_fmt(asked_from)}: interpolation key {_fmt(path)!r} not found")
            if isinstance(node, str) and "${" in node:  # an interpolated node on the way (task_name: ${task.name})
                node = self.value(node, path[: i + 1])
        return self.tree(node, path)

    def _lookup(self, key: str, at: Path) -> Any:
        dots = len(key) - len(key.lstrip("."))
        name = key[dots:]
        if dots:  # omegaconf: one dot = the node's own parent, each further dot one level up
            if dots > len(at):
                raise ConfigError(f"{_fmt(at)}: {key!r} climbs above the root")
            base = at[: len(at) - dots]
        else:
            base = ()
        parts: List[Union[str, int]] = [int(p) if p.isdigit() else p for p in name.split(".")] if name else []
        return self._get(base + tuple(parts), at)

    # -- one ${...}
    def _arg(self, text: str, at: Path) -> Any:
        t = text.strip()
        if len(t) >= 2 and t[0] == t[-1] and t[0] in "\"'":
            return t[1:-1]
        v = self.value(t, at)
        if isinstance(v, str) and v == t:  # a bare literal: typed like a YAML scalar (4096 -> int)
            return yaml.safe_load(t) if t != "" else ""
        return v

    def _eval(self, inner: str, at: Path) -> Any:
        head = inner.split(":", 1)[0]
        if ":" in inner and head and head[0] != "." and head.replace("_", "").replace(".", "").isalnum():
            if head not in RESOLVERS:
                raise ConfigError(f"{_fmt(at)}: unknown resolver {head!r}")
            args = [self._arg(a, at) for a in _split_args(inner.split(":", 1)[1])]
            try:
                return RESOLVERS[head](*args)
            except TypeError as e:
                raise ConfigError(f"{_fmt(at)}: resolver {head!r}: {e}") from None
        return self._lookup(self.value(inner, at) if "${" in inner else inner, at)

    # -- a string value
    def value(self, s: str, at: Path) -> Any:
        if "${" not in s:
            return s
        if at in self._active and s == self._active_text.get(at):
            raise ConfigError(f"{_fmt(at)}: circular interpolation")
        pieces: List[Any] = []
        i = 0
        while i < len(s):
            j = s.find("${", i)
            if j < 0:
                pieces.append(s[i:])
                break
            if j > i:
                pieces.append(s[i:j])
            k = _matching_brace(s, j + 1)
            self._active.append(at)
            self._active_text[at] = s
            try:
                pieces.append(self._eval(s[j + 2:k], at))
            finally:
                self._active.pop()
                self._active_text.pop(at, None)
            i = k + 1
        if len(pieces) == 1:
            return pieces[0]  # a lone interpolation keeps its type
        return "".join(_scalar_str(p) for p in pieces)

    # -- whole tree
    def tree(self, node: Any, at: Path = ()) -> Any:
        if isinstance(node, dict):
            return {k: self.tree(v, at + (k,)) for k, v in node.items()}
        if isinstance(node, list):
            return [self.tree(v, at + (i,)) for i, v in enumerate(node)]
        if isinstance(node, str):
            return self.value(node, at)
        return node


def _scalar_str(v: Any) -> str:
    if isinstance(v, bool):
        return "True" if v else "False"
    return str(v)


def _fmt(path: Sequence[Union[str, int]]) -> str:
    return ".".join(str(p) for p in path) or "<root>"


def resolve(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """All ``${...}`` of an already composed tree resolved; the input is left untouched."""
    return _Resolver(cfg).tree(cfg)


# -------------------------------------------------------------------------------------------------------- front door
def load_config(cfg_dir: str, overrides: Union[None, Dict[str, Any], Iterable[str]] = None, train: Optional[bool] = None
                ) -> Dict[str, Any]:
    """Compose ``cfg_dir/config.yaml`` the way ``@hydra.main(config_name="config", config_path="../cfg")`` does
    (scripts/rlgames_train.py:107) and return the resolved plain dict.

    ``overrides``: hydra command-line items (``["task=Ant", "num_envs=4096", "task.env.episodeLength=500"]``) or the same
    as a dict.  ``task=<Name>`` / ``train=<Name>`` pick group files, every other key is a dotted path into the tree.
    ``train``: load the ``train`` group (rl_games hyper-parameters, not read by the hot path); default: when its file exists.
    """
    ov = _parse_overrides(overrides)
    root = _load_yaml(os.path.join(cfg_dir, "config.yaml"))
    defaults = root.pop("defaults", [])
    root.pop("hydra", None)

    groups: Dict[str, Any] = {}
    for d in defaults:
        if d == "_self_":
            continue
        if isinstance(d, dict):
            for g, name in d.items():
                if g.startswith("override "):  # hydra's own groups (job_logging)
                    continue
                groups[g] = name
        else:
            raise ConfigError(f"config.yaml: unsupported defaults entry {d!r}")
    for g in list(groups):
        if g in ov and not isinstance(ov[g], dict):
            groups[g] = ov.pop(g)

    cfg = dict(root)
    task_name = groups.get("task")
    if task_name is None:
        raise ConfigError("config.yaml: no task group in defaults")
    cfg["task"] = _load_group_file(cfg_dir, "task", str(task_name))
    if "train" in groups:
        train_name = str(groups["train"]).replace("${task}", str(task_name))
        have = os.path.isfile(os.path.join(cfg_dir, "train", train_name + ".yaml"))
        if train is True or (train is None and have):
            cfg["train"] = _load_group_file(cfg_dir, "train", train_name)
    for k, v in ov.items():
        _set_dotted(cfg, k, v)
    if "train" not in cfg:  # wandb_name: ${train.params.config.name} (config.yaml:66) has nothing to point at
        cfg = {k: v for k, v in cfg.items() if not (isinstance(v, str) and "${train." in v)}
    return resolve(cfg)


def load_task_config(cfg_dir: str, task: str, **overrides: Any) -> Dict[str, Any]:
    """``load_config`` without the train group: what ``initialize_task(config, env)`` needs."""
    ov = {"task": task}
    ov.update(overrides)
    return load_config(cfg_dir, ov, train=False)
