"""
RetrievalModelManager — same call surface as coot/model_retrieval.py:57-197 of the reference
(encode_visual / encode_text, model_dict, get_all_params, set_all_models_train/eval,
get/set_model_state), with the four COOT networks running as HIP kernels.

Inside the reference this is selected by the network-type string at coot/model_retrieval.py:80-84:
``name: "transformer_hip"``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from .config import RetrievalConfig, RetrievalNetworksConst, TransformerTypesConst
from .nets import TransformerHip, pack_by_count


@dataclass
class RetrievalDataBatchTuple:
    """Batch layout of coot/dataset_retrieval.py:64-102 (same field names)."""
    key: List[str]
    data_key: List[str]
    sentences: List[List[str]]
    vid_feat: torch.Tensor        # (batch, max_frames, vid_feat_dim) float
    vid_feat_mask: torch.Tensor   # (batch, max_frames) bool, True = padding
    vid_feat_len: torch.Tensor    # (batch) long
    par_feat: torch.Tensor
    par_feat_mask: torch.Tensor
    par_feat_len: torch.Tensor
    clip_num: torch.Tensor        # (batch) long
    clip_feat: torch.Tensor       # (total_clips, max_frames_clip, vid_feat_dim)
    clip_feat_mask: torch.Tensor
    clip_feat_len: torch.Tensor
    sent_num: torch.Tensor
    sent_feat: torch.Tensor
    sent_feat_mask: torch.Tensor
    sent_feat_len: torch.Tensor
    # host-side copies (optional): avoid the device sync of th.max(batch.clip_num) (model_retrieval.py:122)
    max_clip_num: Optional[int] = None
    max_sent_num: Optional[int] = None
    # packed (varlen) token rows of the local networks (attach_packed_index): int32 device row starts [B + Nc + 1]
    # (the B videos / paragraphs first, then the Nc clips / sentences) and their totals on the host.  None: padded processing.
    cu_vis: Optional[torch.Tensor] = None
    cu_txt: Optional[torch.Tensor] = None
    tok_vis: int = 0
    tok_txt: int = 0

    def dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)

    def to_cuda(self, *, non_blocking: bool = True) -> None:
        for name, value in self.__dict__.items():
            if isinstance(value, torch.Tensor):
                setattr(self, name, value.cuda(non_blocking=non_blocking))


@dataclass
class RetrievalPackedBatchTuple:
    """A batch PACKED AT THE SOURCE (dataset_retrieval.collate_fn(packed=True) / coot_collate_packed): the same videos,
    clips, paragraphs and sentences as RetrievalDataBatchTuple (coot/dataset_retrieval.py:64-102) without a single padding row —
    ``vis_tokens`` [tok_vis, vid_feat_dim] holds the frames of the B videos followed by the frames of the Nc clips, ``txt_tokens``
    [tok_txt, text_feat_dim] the words of the B paragraphs followed by those of the Nc sentences (fp32 or bf16), ``cu_vis`` /
    ``cu_txt`` the int32 row starts [B + Nc + 1].  Masks are implied by the lengths.  Consumed by RetrievalTrainer.train_step_native
    (the local networks read the rows in place); ``dataset_retrieval.unpack_batch`` rebuilds the reference's padded batch."""
    key: List[str]
    data_key: List[str]
    sentences: List[List[str]]
    vis_tokens: torch.Tensor
    txt_tokens: torch.Tensor
    cu_vis: torch.Tensor
    cu_txt: torch.Tensor
    vid_feat_len: torch.Tensor
    par_feat_len: torch.Tensor
    clip_num: torch.Tensor
    clip_feat_len: torch.Tensor
    sent_num: torch.Tensor
    sent_feat_len: torch.Tensor
    max_lens: Tuple[int, int, int, int] = (0, 0, 0, 0)   # host: longest video, clip, paragraph, sentence (the padded layout's L)
    max_clip_num: Optional[int] = None
    max_sent_num: Optional[int] = None
    tok_vis: int = 0
    tok_txt: int = 0

    @property
    def device(self) -> torch.device:
        return self.vis_tokens.device

    def dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)


def packed_index(len_a: torch.Tensor, len_b: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """cu_seqlens of two sets of sequences through one local network (set a first): int32 [Na + Nb + 1] exclusive prefix sums of
    the lengths on the lengths' device, and the total as a host int (one device -> host read unless the lengths live on the
    host; loaders compute it from their host-side lengths)."""
    lens = torch.cat([len_a.reshape(-1), len_b.reshape(-1)]).to(torch.int64)
    cu = torch.zeros(lens.numel() + 1, dtype=torch.int32, device=lens.device)
    cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
    return cu, int(cu[-1])


def attach_packed_index(batch: "RetrievalDataBatchTuple", device=None) -> "RetrievalDataBatchTuple":
    """Adds cu_vis / cu_txt / tok_vis / tok_txt to a batch (from host copies of the four length vectors when they are given on the
    host, else with one sync).  The local networks then skip every padding frame / word (coot_packed_seqs)."""
    dev = device if device is not None else batch.vid_feat.device
    batch.cu_vis, batch.tok_vis = packed_index(batch.vid_feat_len.cpu(), batch.clip_feat_len.cpu())
    batch.cu_txt, batch.tok_txt = packed_index(batch.par_feat_len.cpu(), batch.sent_feat_len.cpu())
    batch.cu_vis, batch.cu_txt = batch.cu_vis.to(dev), batch.cu_txt.to(dev)
    return batch


@dataclass
class RetrievalVisualEmbTuple:
    """coot/model_retrieval.py:15-33."""
    vid_emb: torch.Tensor
    clip_emb: torch.Tensor
    vid_context: torch.Tensor
    clip_emb_reshape: torch.Tensor
    clip_emb_mask: torch.Tensor
    clip_emb_lens: torch.Tensor

    def dict(self):
        return dict(self.__dict__)


@dataclass
class RetrievalTextEmbTuple:
    """coot/model_retrieval.py:36-54."""
    par_emb: torch.Tensor
    sent_emb: torch.Tensor
    par_context: torch.Tensor
    sent_emb_reshape: torch.Tensor
    sent_emb_mask: torch.Tensor
    sent_emb_lens: torch.Tensor

    def dict(self):
        return dict(self.__dict__)


class RetrievalModelManager:
    def __init__(self, cfg: RetrievalConfig):
        self.cfg = cfg
        self.model_dict: Dict[str, nn.Module] = {}
        self.was_loaded = False
        self.is_train = True
        for key in RetrievalNetworksConst.values():
            current_cfg = cfg.model_cfgs[key]
            if current_cfg.name in (TransformerTypesConst.TRANSFORMER_HIP, TransformerTypesConst.TRANSFORMER_LEGACY):
                self.model_dict[key] = TransformerHip(current_cfg)
            else:
                raise NotImplementedError(f"Coot model type {current_cfg.name} undefined")
        # data-parallel hook: global max clips / sentences per video over all ranks
        self.global_max_fn = None

    # ---- nntrainer/models/model_manager_base.py:39-163 -----------------------------------------------
    def is_autocast_enabled(self) -> bool:
        return self.cfg.fp16_train if self.is_train else self.cfg.fp16_val

    def get_all_params(self) -> Tuple[Any, Any, Any]:
        params, param_names, params_flat = [], [], []
        wd_bias = getattr(self.cfg.optimizer, "weight_decay_for_bias", False)
        for _model_name, model in self.model_dict.items():
            for key, value in model.named_parameters():
                if not value.requires_grad:
                    continue
                decay_mult = 0.0 if (wd_bias and "bias" in key) else 1.0  # sic: model_manager_base.py:152-154
                params.append({"params": value, "decay_mult": decay_mult, "lr_mult": 1.0})
                param_names.append(key)
                params_flat.append(value)
        return params, param_names, params_flat

    def set_all_models_train(self) -> None:
        self.is_train = True
        for model in self.model_dict.values():
            model.train()

    def set_all_models_eval(self) -> None:
        self.is_train = False
        for model in self.model_dict.values():
            model.eval()

    def get_model_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        return {name: model.state_dict() for name, model in self.model_dict.items()}

    # parameter renames of checkpoints written by the first coot-videotext release (nntrainer/models/model_manager_base.py:96-113)
    LEGACY_RENAMES = {"input_norm.": "norm_input.", "input_fc.": "input_fc.mlp.", "pooler.genpool": "pooler.pools.0.genpool"}

    def set_model_state(self, state) -> None:
        """nntrainer/models/model_manager_base.py:85-126: {network name: state dict} (keys with or without the ``module.``
        prefix of nn.DataParallel), or the LIST of four state dicts of the first coot-videotext release with its parameter names."""
        self.was_loaded = True
        strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        if isinstance(state, (list, tuple)):
            for model_name, this_state in zip(self.model_dict.keys(), state):
                new_state = {}
                for name, param in this_state.items():
                    for old, new in self.LEGACY_RENAMES.items():
                        name = name.replace(old, new)
                    new_state[name] = param
                self.model_dict[model_name].load_state_dict(strip(new_state))
            return
        for model_name, state_dict in state.items():
            self.model_dict[model_name].load_state_dict(strip(state_dict))

    def cuda(self) -> "RetrievalModelManager":
        for k in self.model_dict:
            self.model_dict[k] = self.model_dict[k].cuda()
        return self

    def mark_weights_dirty(self) -> None:
        for m in self.model_dict.values():
            m.mark_dirty()

    # ---- coot/model_retrieval.py:86-197 --------------------------------------------------------------------
    def _encode(self, net_local, net_global, ctx_feat, ctx_mask, ctx_len, item_feat, item_mask, item_len, item_num,
                cmax: Optional[int], packed=None):
        # both passes through the local network share one call (same weights): every GEMM / LayerNorm launch
        # covers the video-level AND the clip-level tokens
        context, item_emb = net_local.forward_pair(ctx_feat, ctx_len, item_feat, item_len, packed=packed)
        if cmax is None:
            cmax = int(torch.max(item_num))
        if self.global_max_fn is not None:
            cmax = int(self.global_max_fn(cmax))
        reshape, mask, lens = pack_by_count(item_emb, item_num, cmax)
        hidden = context if net_global.cfg.use_context else None
        glob, _ = net_global(reshape, mask, item_num, hidden, want_tokens=False)
        return glob, item_emb, context, reshape, mask, lens

    def encode_visual(self, batch: RetrievalDataBatchTuple) -> RetrievalVisualEmbTuple:
        K = RetrievalNetworksConst
        out = self._encode(self.model_dict[K.NET_VIDEO_LOCAL], self.model_dict[K.NET_VIDEO_GLOBAL], batch.vid_feat,
                           batch.vid_feat_mask, batch.vid_feat_len, batch.clip_feat, batch.clip_feat_mask,
                           batch.clip_feat_len, batch.clip_num, getattr(batch, "max_clip_num", None),
                           (batch.cu_vis, batch.tok_vis) if getattr(batch, "cu_vis", None) is not None else None)
        return RetrievalVisualEmbTuple(*out)

    def encode_text(self, batch: RetrievalDataBatchTuple) -> RetrievalTextEmbTuple:
        K = RetrievalNetworksConst
        out = self._encode(self.model_dict[K.NET_TEXT_LOCAL], self.model_dict[K.NET_TEXT_GLOBAL], batch.par_feat,
                           batch.par_feat_mask, batch.par_feat_len, batch.sent_feat, batch.sent_feat_mask,
                           batch.sent_feat_len, batch.sent_num, getattr(batch, "max_sent_num", None),
                           (batch.cu_txt, batch.tok_txt) if getattr(batch, "cu_txt", None) is not None else None)
        return RetrievalTextEmbTuple(*out)
