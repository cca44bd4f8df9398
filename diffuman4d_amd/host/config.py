"""Config composition for the CLI: Hydra when it is installed, a small built-in composer otherwise.

The reference's entry point is ``python inference.py exp=<name> data.scene_label=... ...`` with a Hydra
1.3 config tree (``/root/reference/configs``: root ``test.yaml`` + groups ``data/ model/ sampler/ exp/``,
``_target_`` instantiation).  Neither hydra-core nor omegaconf is available in the
build image, so this module implements the subset of the grammar those configs use:

  * group selection ``group=name`` and dotted value overrides ``a.b.c=value`` (YAML-typed values)
  * ``defaults:`` lists inside group files (``- base``, ``- _self_``, ``- override /group: name``,
    ``- .: sibling``) and ``# @package _global_`` experiment overlays
  * ``${a.b}`` interpolation, ``${oc.env:VAR,default}``, ``${now:fmt}``, ``${hydra:runtime.choices.exp}``
  * ``_target_`` instantiation with keyword overrides

The defaults below restate the VALUES of the reference's config groups (they are part of the drop-in
surface); a user can instead point ``--config-dir`` at the reference's own ``configs/`` directory, whose
``_target_`` strings are mapped onto this package by ``TARGET_ALIASES``.
"""
from __future__ import annotations

import copy
import importlib
import os
import re
import time
from pathlib import Path
from typing import Any, Dict, List, Optional

import yaml

_DATA_PATS = {
    "camera_path_pat": "{data_dir}/{scene_label}/transforms.json",
    "image_path_pat": "{data_dir}/{scene_label}/images/{spa_label}/{tem_label}%s",
    "fmask_path_pat": "{data_dir}/{scene_label}/fmasks/{spa_label}/{tem_label}.png",
    "skeleton_path_pat": "{data_dir}/{scene_label}/skeletons/{spa_label}/{tem_label}%s",
}


def _data(data_dir: str, ext: str) -> Dict[str, Any]:
    d = {"_target_": "src.data.spatem_dataset.SpaTemDataset", "data_dir": data_dir}
    d.update({k: (v % ext if "%s" in v else v) for k, v in _DATA_PATS.items()})
    d.update({"scene_label": "0326_07", "has_gt_target": True})
    return d


_SLIDING_DEFAULT = {
    "_target_": "src.samplers.sliding_iterative_sampler.SlidingIterativeSampler",
    "output_dir": "${result_dir}/${data.scene_label}",
    "window_size": 12, "sliding_stride": 1, "sliding_shift": 0, "bidirectional": False,
    "num_denoising_steps": 1, "alternation_rounds": 3, "guidance_scale": 2.0,
    "spa_label_range": [0, 48, 1], "tem_label_range": [0, 150, 1], "spa_labels": None, "tem_labels": None,
    "input_spa_labels": [1, 13, 25, 37],
}
_RANGES = {"spa_label_range": [0, 48, 1], "input_spa_labels": [1, 13, 25, 37]}

# group -> name -> (defaults list, body); same group / option names and values as the reference tree
BUILTIN: Dict[str, Dict[str, Any]] = {
    "data": {
        "dna_rendering": {"body": _data("./data/dna_rendering_processed", ".webp")},
        "fdvai": {"body": _data("./data/fdvai", ".jpg")},
        "synthetic": {"body": {"_target_": "diffuman4d_amd.host.dataset.SyntheticSpaTemDataset", "scene_label": "synthetic",
                               "height": 576, "width": 320, "num_cameras": 48}},
    },
    "model": {
        "diffuman4d": {"body": {"_target_": "src.samplers.utils.sampling_utils.load_pipelines", "repo_id": "krahets/Diffuman4D",
                                "model_dir": "./models/models--krahets--Diffuman4D", "torch_dtype": "bf16", "gpu_ids": None}},
        "diffuman4d_mi355x": {"body": {"_target_": "diffuman4d_amd.host.loader.load_pipelines", "repo_id": "krahets/Diffuman4D",
                                       "model_dir": "./models/models--krahets--Diffuman4D", "torch_dtype": "bf16",
                                       "gpu_ids": None, "precision": "auto"}},
    },
    "sampler": {
        "sliding_default": {"body": _SLIDING_DEFAULT},
        "sliding_fast": {"defaults": ["sliding_default", "_self_"], "body": {"sliding_stride": 2}},
        "sliding_3d": {"defaults": ["sliding_default", "_self_"], "body": {"alternation_rounds": 1}},
        "sliding_premium": {"defaults": ["sliding_default", "_self_"], "body": {"alternation_rounds": 5}},
        "sliding_low_mem": {"defaults": ["sliding_default", "_self_"], "body": {"window_size": 4, "guidance_scale": 1.0}},
    },
    "exp": {
        "demo_4d": {"global": True,
                    "defaults": [{"override /data": "dna_rendering"}, {"override /model": "diffuman4d"},
                                 {"override /sampler": "sliding_fast"}],
                    "body": {"data": {"scene_label": "0811_06", "has_gt_target": True},
                             "model": {"torch_dtype": "bf16", "gpu_ids": None},
                             "sampler": dict(_RANGES, tem_label_range=[0, 150, 1]),
                             "sampling": True, "to_nerfstudio": True, "evaluating": False}},
        "demo_4d_tiny": {"global": True, "defaults": [{".": "demo_4d"}],
                         "body": {"sampler": dict(_RANGES, tem_label_range=[0, 16, 1])}},
        "demo_3d": {"global": True, "defaults": [{".": "demo_4d"}, {"override /sampler": "sliding_3d"}],
                    "body": {"sampler": dict(_RANGES, tem_label_range=[0, 1, 1])}},
    },
}
ROOT_DEFAULTS = {"data": "dna_rendering", "model": "diffuman4d", "sampler": "sliding_default"}
ROOT_BODY = {
    "exp_name": "${hydra:runtime.choices.exp}",
    "timestamp": "${oc.env:TIMESTAMP,${now:%Y%m%d_%H%M%S}}",
    "log_dir": "./output/logs/${exp_name}/${timestamp}",
    "result_dir": "./output/results/${exp_name}",
    "sampling": True, "to_nerfstudio": True, "evaluating": False,
}

# reference `_target_` strings -> this package (a reference checkout on sys.path wins for the dataset)
TARGET_ALIASES = {
    "src.samplers.sliding_iterative_sampler.SlidingIterativeSampler": "diffuman4d_amd.host.sampler.SlidingIterativeSampler",
    "src.samplers.utils.sampling_utils.load_pipelines": "diffuman4d_amd.host.loader.load_pipelines",
}


def _merge(dst: Dict, src: Dict) -> Dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


class _Source:
    """Group files from a Hydra-style directory, falling back to BUILTIN."""

    def __init__(self, config_dir: Optional[str]):
        self.dir = Path(config_dir) if config_dir else None

    def load(self, group: str, name: str) -> Dict[str, Any]:
        if self.dir is not None and (self.dir / group / f"{name}.yaml").exists():
            text = (self.dir / group / f"{name}.yaml").read_text()
            body = yaml.safe_load(text) or {}
            defaults = body.pop("defaults", [])
            return {"defaults": defaults, "body": body, "global": "@package _global_" in text.split("\n", 1)[0]}
        if group in BUILTIN and name in BUILTIN[group]:
            e = BUILTIN[group][name]
            return {"defaults": e.get("defaults", []), "body": copy.deepcopy(e["body"]), "global": e.get("global", False)}
        raise KeyError(f"config group option '{group}/{name}' not found")

    def resolve_group(self, group: str, name: str, choices: Dict[str, str]) -> Dict[str, Any]:
        """Body of group/name with its own defaults list applied (group-local inheritance + overrides)."""
        e = self.load(group, name)
        out: Dict[str, Any] = {}
        self_done = False
        for d in e["defaults"]:
            if d == "_self_":
                _merge(out, e["body"])
                self_done = True
            elif isinstance(d, str):
                _merge(out, self.resolve_group(group, d, choices))
            elif isinstance(d, dict):
                (k, v), = d.items()
                if k == ".":
                    _merge(out, self.resolve_group(group, v, choices))
                elif k.startswith("override /"):
                    choices[k[len("override /"):]] = v
                else:
                    raise ValueError(f"unsupported defaults entry {d!r} in {group}/{name}")
        if not self_done:
            _merge(out, e["body"])
        return out


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(cfg: Dict, path: str):
    cur: Any = cfg
    for p in path.split("."):
        cur = cur[p]
    return cur


def _resolve(cfg: Dict, choices: Dict[str, str]) -> Dict:
    now = time.localtime()

    def sub(s: str, depth=0) -> Any:
        if depth > 16:
            raise ValueError(f"interpolation too deep: {s}")
        while True:
            m = _INTERP.search(s)
            if not m:
                return s
            key = m.group(1)
            if key.startswith("oc.env:"):
                var, _, default = key[len("oc.env:"):].partition(",")
                val = os.environ.get(var, default)
            elif key.startswith("now:"):
                val = time.strftime(key[len("now:"):], now)
            elif key.startswith("hydra:runtime.choices."):
                val = choices[key[len("hydra:runtime.choices."):]]
            elif key.startswith("hydra:"):
                val = ""
            else:
                val = _lookup(cfg, key)
                if isinstance(val, str):
                    val = sub(val, depth + 1)
            if m.start() == 0 and m.end() == len(s):
                return val
            s = s[: m.start()] + str(val) + s[m.end():]

    def walk(node):
        if isinstance(node, dict):
            return {k: walk(v) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v) for v in node]
        if isinstance(node, str):
            return sub(node)
        return node

    return walk(cfg)


_INT = re.compile(r"^[+-]?(0|[1-9][0-9]*)$")


def _parse_value(v: str):
    """Hydra override values: ints without leading zeros, floats, booleans, null, [lists]; anything
    else (e.g. the scene label 0023_06) stays a string -- YAML 1.1 would read that as an octal int."""
    s = v.strip()
    if s[:1] in "[{":
        return yaml.safe_load(s)
    if (s[:1] == s[-1:]) and s[:1] in "'\"" and len(s) >= 2:
        return s[1:-1]
    if _INT.match(s):
        return int(s)
    low = s.lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none", "~"):
        return None
    try:
        if any(c in s for c in ".eE") and not s[:2].lstrip("+-").startswith("0") or s.startswith(("0.", "-0.", "+0.")):
            return float(s)
    except ValueError:
        pass
    return s


def compose(overrides: List[str], config_dir: Optional[str] = None) -> Dict[str, Any]:
    """Hydra-like composition: returns a plain, fully resolved nested dict."""
    src = _Source(config_dir)
    choices = dict(ROOT_DEFAULTS)
    values: List[tuple] = []
    for ov in overrides:
        k, eq, v = ov.partition("=")
        if not eq:
            raise ValueError(f"override '{ov}' must look like key=value")
        k = k.lstrip("+")
        if "." not in k and (k in BUILTIN or k == "exp"):
            choices[k] = v
        else:
            values.append((k, _parse_value(v)))
    if "exp" not in choices:
        raise ValueError("You must specify 'exp', e.g. exp=demo_4d")
    cli_choices = {k: v for k, v in choices.items()}
    exp_body = src.resolve_group("exp", choices["exp"], choices)  # may override group choices
    for k, v in cli_choices.items():  # command-line group choices win over the experiment's
        if k != "exp" and any(o.startswith(k + "=") for o in overrides):
            choices[k] = v
    cfg: Dict[str, Any] = copy.deepcopy(ROOT_BODY)
    if config_dir and (Path(config_dir) / "test.yaml").exists():
        root = yaml.safe_load((Path(config_dir) / "test.yaml").read_text()) or {}
        root.pop("defaults", None)
        cfg = root
    for g in ("data", "model", "sampler"):
        cfg[g] = src.resolve_group(g, choices[g], choices)
    _merge(cfg, exp_body)
    for k, v in values:
        cur = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = v
    return _resolve(cfg, choices)


def locate(path: str):
    path = TARGET_ALIASES.get(path, path)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Dict[str, Any], **kwargs):
    """hydra.utils.instantiate for flat ``_target_`` nodes."""
    node = dict(node)
    target = locate(node.pop("_target_"))
    node.update(kwargs)
    return target(**node)
