"""Plumbing-only stand-in for the `diffusers` package (absent from this image, no network).

Used ONLY by tools/oracle/gen_golden.py in the build container to import the reference's
in-tree modules (`/root/reference/magicanimate/models/*.py`).  It contains NO arithmetic:
every FLOP executed through it is reference-authored code, because the attention /
feed-forward / embedding classes that the reference imports from `diffusers` are aliased
to the reference's *own* in-tree copies (`magicanimate/models/orig_attention.py`,
`magicanimate/models/embeddings.py`)  -- SURVEY.md Appendix A.

Nothing here travels to the GPU box and nothing here is imported by the product.
"""
from __future__ import annotations

import functools
import inspect
import logging as _pylogging
import sys
import types
from collections import OrderedDict

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kw):
        cfg = getattr(self, "_internal_dict", None)
        if cfg is None:
            cfg = _AttrDict()
            object.__setattr__(self, "_internal_dict", cfg)
        cfg.update(kw)

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kw):
        sig = inspect.signature(cls.__init__).parameters
        merged = {k: v for k, v in dict(config).items() if k in sig}
        merged.update({k: v for k, v in kw.items() if k in sig})
        return cls(**merged)


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        ConfigMixin.register_to_config(self, **cfg)

    return wrapper


class BaseOutput(OrderedDict):
    """dataclass-style output: supports .field, ["field"], [0]."""

    def __post_init__(self):
        import dataclasses

        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return list(self.values())[k]


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    """Register fake `diffusers.*` modules, then alias the arithmetic classes to the
    reference's in-tree ones.  Idempotent."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_emo_shim", False):
        return
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True

    d = _mod("diffusers")
    d._emo_shim = True
    d.__path__ = []
    cu = _mod("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config, cu.FrozenDict = ConfigMixin, register_to_config, _AttrDict
    u = _mod("diffusers.utils")
    u.__path__ = []
    u.BaseOutput = BaseOutput
    u.WEIGHTS_NAME = "diffusion_pytorch_model.bin"
    lg = types.SimpleNamespace(get_logger=lambda name=None: _pylogging.getLogger(name or "diffusers"))
    u.logging = lg
    iu = _mod("diffusers.utils.import_utils")
    iu.is_xformers_available = lambda: False
    u.import_utils = iu
    m = _mod("diffusers.models")
    m.__path__ = []
    mu = _mod("diffusers.models.modeling_utils")
    mu.ModelMixin = ModelMixin
    m.ModelMixin = ModelMixin

    # embeddings -> reference's own in-tree embeddings.py (torch/numpy only)
    import magicanimate.models.embeddings as ref_emb

    e = _mod("diffusers.models.embeddings")
    for n in ("TimestepEmbedding", "Timesteps", "ImagePositionalEmbeddings", "GaussianFourierProjection"):
        if hasattr(ref_emb, n):
            setattr(e, n, getattr(ref_emb, n))

    # attention -> reference's own in-tree orig_attention.py
    import magicanimate.models.orig_attention as ref_att

    a = _mod("diffusers.models.attention")
    a.FeedForward = ref_att.FeedForward
    a.AdaLayerNorm = ref_att.AdaLayerNorm
    a.BasicTransformerBlock = ref_att.BasicTransformerBlock
    a.Attention = ref_att.CrossAttention
    a.CrossAttention = ref_att.CrossAttention

    # isinstance-only classes for the (disabled) AdaIN path of mutual_self_attention
    b = _mod("diffusers.models.unet_2d_blocks")
    for n in ("CrossAttnDownBlock2D", "CrossAttnUpBlock2D", "DownBlock2D", "UpBlock2D"):
        setattr(b, n, type(n, (nn.Module,), {}))

    # torch_dfs: extracted from the reference at run time (nothing re-typed into the repo)
    import ast

    path = f"{REFERENCE_ROOT}/magicanimate/models/stable_diffusion_controlnet_reference.py"
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "torch_dfs"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    s = _mod("magicanimate.models.stable_diffusion_controlnet_reference")
    s.torch_dfs = ns["torch_dfs"]


def extract_classes(pyfile, names, extra_ns=None):
    """AST-extract the FIRST ClassDef of each name from a reference file whose module-level
    imports fail here (Net.py, train_stage_*.py) and exec just those class bodies."""
    import ast
    import math
    import typing

    import torch.nn.functional as F

    tree = ast.parse(open(pyfile).read())
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "np": __import__("numpy"),
          "ModelMixin": ModelMixin, "Optional": typing.Optional, "List": typing.List,
          "Tuple": typing.Tuple, "Union": typing.Union, "Dict": typing.Dict, "Any": typing.Any}
    ns.update(extra_ns or {})
    seen = set()
    body = []
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name in names and n.name not in seen:
            seen.add(n.name)
            body.append(n)
    exec(compile(ast.Module(body=body, type_ignores=[]), pyfile, "exec"), ns)
    return {k: ns[k] for k in names if k in ns}
