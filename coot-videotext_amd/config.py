"""
Experiment configuration for the retrieval hot path.

Reads the reference's YAML schema unchanged (config/retrieval/paper2020/*.yaml of the reference:
same key names, `same_as` inheritance as in nntrainer/utils.py:220-256) so an existing experiment
file works as-is; only the fields the hot path consumes are interpreted
(coot/configs_retrieval.py:14-189, nntrainer/models/transformer_legacy.py:26-97).
The idiomatic switch to this implementation inside the reference is the network-type string
dispatched at coot/model_retrieval.py:80-84: name "transformer" (reference) vs "transformer_hip".
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional

import yaml

from . import lib as _lib


class RetrievalNetworksConst:
    """coot/configs_retrieval.py:182-189."""
    NET_VIDEO_LOCAL = "net_video_local"
    NET_VIDEO_GLOBAL = "net_video_global"
    NET_TEXT_LOCAL = "net_text_local"
    NET_TEXT_GLOBAL = "net_text_global"

    @classmethod
    def values(cls):
        return [cls.NET_VIDEO_LOCAL, cls.NET_VIDEO_GLOBAL, cls.NET_TEXT_LOCAL, cls.NET_TEXT_GLOBAL]


class TransformerTypesConst:
    """nntrainer/models/transformer_legacy.py:100-110 plus the HIP type."""
    TRANSFORMER_LEGACY = "transformer"
    TRANSFORMER_HIP = "transformer_hip"


def resolve_same_as(config: Dict[str, Any], root: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """`same_as: other_key` deep-copies other_key's dict, then applies the local overrides
    (nntrainer/utils.py:220-256)."""
    root = config if root is None else root
    for key, val in list(config.items()):
        if isinstance(val, dict):
            if "same_as" in val:
                base = copy.deepcopy(resolve_same_as({val["same_as"]: root[val["same_as"]]}, root)[val["same_as"]])
                over = {k: v for k, v in val.items() if k != "same_as"}
                base.update(over)
                config[key] = base
            resolve_same_as(config[key], root)
    return config


class _Section:
    def __init__(self, d: Dict[str, Any]):
        for k, v in d.items():
            setattr(self, k, _Section(v) if isinstance(v, dict) else v)

    def dict(self):
        return {k: (v.dict() if isinstance(v, _Section) else v) for k, v in self.__dict__.items()}


class TransformerConfig:
    """Fields of one net_* section that the hot path reads."""

    def __init__(self, d: Dict[str, Any], input_dim: int):
        self.name: str = d.get("name", TransformerTypesConst.TRANSFORMER_LEGACY)
        self.output_dim: int = d["output_dim"]
        self.input_dim = input_dim
        sa = d["selfatn_config"]
        self.hidden_dim: int = sa["hidden_dim"]
        self.num_layers: int = sa["num_layers"]
        self.num_heads: int = sa["num_heads"]
        self.ff_dim: int = sa["pointwise_ff_dim"] or sa["hidden_dim"]
        self.dropout: float = float(sa["dropout"])
        self.use_input_fc: bool = bool(d["use_input_fc"])
        self.use_context: bool = bool(d["use_context"])
        self.ctx_num_layers = 1
        self.ctx_dropout = 0.0
        if self.use_context:
            ca = d["crossatn_config"]
            self.ctx_num_layers = ca["num_layers"]
            self.ctx_dropout = float(ca["dropout"])
            _require(ca["hidden_dim"] == self.hidden_dim and ca["num_heads"] == self.num_heads and
                     (ca["pointwise_ff_dim"] or ca["hidden_dim"]) == self.ff_dim,
                     "crossatn_config must match selfatn_config dims")
        pc = d["pooler_config"]
        self.pooler: str = pc["name"]
        self.pool_hidden: int = pc.get("hidden_dim", 0) or self.hidden_dim
        self.pool_heads: int = pc.get("num_heads", 1)
        self.pool_dropout: float = float(pc.get("dropout", 0))
        # compute mode of the HIP network (not a reference key): "bf16" = the fast path, "f32" = the fp32 reference
        # mode (include/coot_hip.h: coot_net_config.dtype), e.g. to tell bf16 rounding from a logic error
        # "f16": IEEE half operands = the second build of the library (COOT_OPERAND=f16 -> libcoot_hip_f16.so, forward-only); the default
        # is the format of the library the process selected, and a configuration that names the other one is refused by the library
        self.dtype: str = d.get("dtype", _lib.OPERAND_ENV)
        _require(self.dtype in ("bf16", "f16", "f32"), f"dtype {self.dtype} (bf16, f16 or f32)")
        self.weight_init_type: str = d.get("weight_init_type", "truncnorm")
        self.weight_init_std: float = d.get("weight_init_std", 0.01)
        # options of the reference that no shipped config enables and the HIP path does not implement
        _require(self.pooler in ("atn", "avg_special"), f"pooler {self.pooler} not supported")
        _require(not d.get("add_local_cls_token", False), "add_local_cls_token not supported")
        _require(not d.get("use_output_fc", False), "use_output_fc not supported")
        _require(not d.get("linear_out", False), "linear_out not supported")
        _require(d.get("dropout_input", 0) == 0, "dropout_input not supported")
        _require(d.get("norm_input", "layernorm_coot") == "layernorm_coot", "norm_input must be layernorm_coot")
        _require(d.get("positional_encoding", "sincos") == "sincos", "positional_encoding must be sincos")
        _require(sa.get("norm", "layernorm_coot") in ("layernorm_coot", {"name": "layernorm_coot"}), "norm must be layernorm_coot")
        _require(sa.get("activation", "gelu") == "gelu", "activation must be gelu")
        if self.use_input_fc:
            fc = d["input_fc_config"]
            _require(fc["num_layers"] == 1 and fc["activation_output"] == "gelu" and fc["output_dim"] == self.hidden_dim
                     and fc.get("residual", "none") == "none" and fc.get("norm_output", "none") == "none"
                     and not fc.get("dropout_output", 0), "input_fc_config: only 1-layer Linear+GELU is supported")
        if self.pooler == "atn":
            _require(pc.get("num_layers", 1) == 1 and pc.get("activation", "gelu") == "gelu", "atn pooler: 1 pool, gelu")

    def to_c(self) -> _lib.NetConfig:
        return _lib.NetConfig(self.input_dim, self.hidden_dim, self.num_heads, self.ff_dim, self.num_layers,
                              int(self.use_input_fc), int(self.use_context), self.ctx_num_layers,
                              0 if self.pooler == "atn" else 1, self.pool_hidden, self.pool_heads,
                              self.dropout, self.ctx_dropout, self.pool_dropout, {"bf16": _lib.DTYPE_BF16, "f32": _lib.DTYPE_F32, "f16": _lib.DTYPE_F16}[self.dtype])


def _require(cond: bool, msg: str):
    if not cond:
        raise NotImplementedError(f"coot HIP path: {msg}")


class RetrievalConfig:
    """Mirror of coot/configs_retrieval.py:14-54 (fields used by model manager and loss hooks)."""

    def __init__(self, config: Dict[str, Any]):
        config = resolve_same_as(copy.deepcopy(config))
        self.raw = config
        self.train = _Section(config["train"])
        self.val = _Section(config.get("val", {}))
        self.dataset_train = _Section(config.get("dataset_train", {}))
        self.dataset_val = _Section(config.get("dataset_val", config.get("dataset_train", {})))
        self.optimizer = _Section(config.get("optimizer", {}))
        self.use_cuda: bool = config.get("use_cuda", True)
        self.fp16_train: bool = config.get("fp16_train", True)
        self.fp16_val: bool = config.get("fp16_val", True)
        vid_dim, txt_dim = self.dataset_val.vid_feat_dim, self.dataset_val.text_feat_dim
        K = RetrievalNetworksConst
        input_dims = {K.NET_VIDEO_LOCAL: vid_dim, K.NET_VIDEO_GLOBAL: config[K.NET_VIDEO_LOCAL]["output_dim"],
                      K.NET_TEXT_LOCAL: txt_dim, K.NET_TEXT_GLOBAL: config[K.NET_TEXT_LOCAL]["output_dim"]}
        self.model_cfgs: Dict[str, TransformerConfig] = {k: TransformerConfig(config[k], input_dims[k]) for k in K.values()}


def load_yaml_config_file(path: str) -> Dict[str, Any]:
    with open(path, "rt", encoding="utf8") as fh:
        return yaml.safe_load(fh)


CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "retrieval")


def load_named_config(name: str, vid_feat_dim: Optional[int] = None) -> RetrievalConfig:
    """name in {anet_coot, yc2_100m_coot, yc2_2d3d_coot} (config/retrieval/*.yaml in this repo); vid_feat_dim overrides the
    video feature width of the dataset sections (the reference's `-o dataset_train.vid_feat_dim=...`)."""
    d = load_yaml_config_file(os.path.join(CONFIG_DIR, name + ".yaml"))
    if vid_feat_dim is not None:
        d = resolve_same_as(d)
        for sec in ("dataset_train", "dataset_val"):
            if sec in d:
                d[sec]["vid_feat_dim"] = int(vid_feat_dim)
    return RetrievalConfig(d)
