"""
Losses of the retrieval hot path as HIP kernels behind the reference's interfaces
(coot/loss_fn.py: ContrastiveLoss :51-100, CycleConsistencyLoss :111-387;
coot/trainer_retrieval.py:148-233 hooks).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as _lib


class ContrastiveLossConfig:
    """coot/loss_fn.py:33-48."""

    def __init__(self, margin=0.2, weight_high=1.0, weight_high_internal=1.0, weight_low=1.0, weight_low_internal=1.0,
                 weight_context=1.0, weight_context_internal=0.0):
        self.margin, self.weight_high, self.weight_high_internal = margin, weight_high, weight_high_internal
        self.weight_low, self.weight_low_internal = weight_low, weight_low_internal
        self.weight_context, self.weight_context_internal = weight_context, weight_context_internal

    @classmethod
    def from_section(cls, sec) -> "ContrastiveLossConfig":
        d = sec.dict() if hasattr(sec, "dict") else dict(sec)
        return cls(**{k: float(d[k]) for k in ("margin", "weight_high", "weight_high_internal", "weight_low",
                                                "weight_low_internal", "weight_context", "weight_context_internal")})

    def to_c(self) -> _lib.ContrastiveConfig:
        return _lib.ContrastiveConfig(self.margin, self.weight_high, self.weight_high_internal, self.weight_low,
                                      self.weight_low_internal, self.weight_context, self.weight_context_internal)


class _TotalContrastiveFn(torch.autograd.Function):
    """compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182): F.normalize of the six
    embedding sets + 3 alignment + 4 clustering ContrastiveLoss terms; forward and gradient are produced
    by ONE C call (the backward is computed eagerly into saved buffers)."""

    @staticmethod
    def forward(ctx, cfg: ContrastiveLossConfig, vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx, dtype: str = "bf16"):
        lib = _lib.load()
        assert dtype in ("bf16", "f32"), dtype
        embs = [t.contiguous().float() for t in (vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx)]
        n_high, d_high = embs[0].shape
        n_low, d_low = embs[2].shape
        assert embs[1].shape == embs[0].shape and embs[3].shape == embs[2].shape
        assert embs[4].shape == (n_high, d_low) and embs[5].shape == (n_high, d_low), "context dims"
        dev = embs[0].device
        need_grad = any(ctx.needs_input_grad[1:])
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        grads = [torch.zeros_like(e) for e in embs] if need_grad else [None] * 6
        # "f32": the fp32 reference mode of the loss (coot_contrastive_fwd_bwd_f32: plain FMA kernels, no bf16 operand) — the loss-side
        # counterpart of TransformerHip.set_compute_dtype("f32"); a checker, not a fast path
        size_fn, call, name = ((lib.coot_contrastive_f32_scratch_bytes, lib.coot_contrastive_fwd_bwd_f32, "coot_contrastive_fwd_bwd_f32")
                               if dtype == "f32" else
                               (lib.coot_contrastive_scratch_bytes, lib.coot_contrastive_fwd_bwd, "coot_contrastive_fwd_bwd"))
        scratch = torch.empty(size_fn(n_high, n_low, d_high, d_low), dtype=torch.uint8, device=dev)
        ccfg = cfg.to_c()
        _lib.check(call(C.byref(ccfg), n_high, n_low, d_high, d_low, *[_lib.ptr(e) for e in embs],
                        _lib.ptr(loss), *[_lib.ptr(g) for g in grads], _lib.ptr(scratch),
                        scratch.numel(), _lib.stream_ptr()), name)
        ctx.grads = grads
        return loss

    @staticmethod
    def backward(ctx, dloss):
        return (None,) + tuple(None if g is None else g * dloss for g in ctx.grads) + (None,)


def total_contrastive_loss(cfg: ContrastiveLossConfig, vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx, dtype: str = "bf16"):
    """dtype "f32": the library's fp32 reference mode of the loss (include/coot_hip.h: coot_contrastive_fwd_bwd_f32)."""
    return _TotalContrastiveFn.apply(cfg, vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx, dtype)


class ContrastiveLoss(torch.nn.Module):
    """ContrastiveLoss(margin)(im, s) (coot/loss_fn.py:51-100).  The reference's training setting — max_violation=False,
    norm=True, the only one its trainer constructs (coot/trainer_retrieval.py:84-86) — runs on the fused HIP loss; the two
    constructor flags no configuration reaches (hardest negative only, un-normalised sum) are a few tensor ops on the
    embeddings' device, differentiated by autograd."""

    def __init__(self, margin: float, max_violation: bool = False, norm: bool = True, use_cuda: bool = True):
        super().__init__()
        self.margin, self.max_violation, self.norm = margin, bool(max_violation), bool(norm)

    def _general(self, im: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
        scores = im @ s.t()   # cosine_sim (coot/loss_fn.py:19-30): the inputs are L2-normalised by the caller
        diag = scores.diag()
        off = ~torch.eye(scores.shape[0], dtype=torch.bool, device=scores.device)
        cost_s = (self.margin + scores - diag[:, None]).clamp(min=0) * off     # caption retrieval: against the row's own pair
        cost_im = (self.margin + scores - diag[None, :]).clamp(min=0) * off    # image retrieval: against the column's own pair
        if self.max_violation:                                                 # the hardest negative of every query only
            cost_s, cost_im = cost_s.max(1)[0], cost_im.max(0)[0]
        total = cost_s.sum() + cost_im.sum()
        return total / (im.shape[0] * s.shape[0]) if self.norm else total

    def forward(self, im: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
        if self.max_violation or not self.norm:
            return self._general(im, s)
        # one alignment term; inputs are (already) normalised, re-normalising is the identity
        cfg = ContrastiveLossConfig(self.margin, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        dummy = im.new_zeros((8, 8))
        ctxd = im.new_zeros((im.shape[0], 8))
        return _TotalContrastiveFn.apply(cfg, im, s, dummy, dummy, ctxd, ctxd)


# Deterministic mode (RetrievalTrainer.set_deterministic): the cycle-consistency loss word is the sum of 2 B atomic addends.  A
# persistent word the trainer registered with the library's fixed-point accumulators (coot_det_configure) takes them; it is flushed
# and copied out right here, so the returned loss is bit-reproducible like the gradients.
_DET_LOSS_WORD: Optional[torch.Tensor] = None


def set_det_loss_word(word: Optional[torch.Tensor]) -> None:
    global _DET_LOSS_WORD
    _DET_LOSS_WORD = word


class _CycleConsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, sent, clip_lens, sent_lens, idx_clip, idx_sent, weight: float, inv_batch: float, want_rows: bool):
        lib = _lib.load()
        clip, sent = clip.contiguous().float(), sent.contiguous().float()
        B, Cc, D = clip.shape
        Cs = sent.shape[1]
        dev = clip.device
        word = _DET_LOSS_WORD if (_DET_LOSS_WORD is not None and _DET_LOSS_WORD.device == dev) else None
        if word is not None:
            word.zero_()
        loss = word if word is not None else torch.zeros((), dtype=torch.float32, device=dev)
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dclip = torch.zeros_like(clip) if need_grad else None
        dsent = torch.zeros_like(sent) if need_grad else None
        rows_c = torch.empty(B, Cc, dtype=torch.float32, device=dev) if want_rows else None
        rows_s = torch.empty(B, Cs, dtype=torch.float32, device=dev) if want_rows else None
        _lib.check(lib.coot_cyclecons_fwd_bwd(_lib.ptr(clip), _lib.ptr(sent), _lib.ptr(clip_lens.contiguous().long()),
                                              _lib.ptr(sent_lens.contiguous().long()), _lib.ptr(idx_clip.contiguous().long()),
                                              _lib.ptr(idx_sent.contiguous().long()), B, Cc, Cs, D, float(weight), float(inv_batch),
                                              _lib.ptr(loss), _lib.ptr(rows_c), _lib.ptr(rows_s), _lib.ptr(dclip), _lib.ptr(dsent),
                                              _lib.stream_ptr()), "coot_cyclecons_fwd_bwd")
        if word is not None:
            _lib.check(lib.coot_det_flush(word.data_ptr(), 4, _lib.stream_ptr()), "coot_det_flush")
            loss = word.clone().reshape(())
        ctx.grads = (dclip, dsent)
        ctx.mark_non_differentiable(*[r for r in (rows_c, rows_s) if r is not None])
        return loss, rows_c, rows_s

    @staticmethod
    def backward(ctx, dloss, _a, _b):
        dclip, dsent = ctx.grads
        return (None if dclip is None else dclip * dloss, None if dsent is None else dsent * dloss) + (None,) * 7


def sample_cycle_indices(lens: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """One uniformly random valid position per video — th.multinomial(mask.float(), 1) of
    coot/loss_fn.py:306-314, drawn for the whole batch at once on the device."""
    u = torch.rand(lens.shape, device=lens.device, generator=generator)
    return torch.minimum((u * lens.float()).long(), lens.long() - 1)


def cycle_consistency_loss(clip_emb_reshape, clip_lens, sent_emb_reshape, sent_lens, weight: float,
                           idx_clip: Optional[torch.Tensor] = None, idx_sent: Optional[torch.Tensor] = None,
                           global_batch: Optional[int] = None, want_rows: bool = False):
    """weight * (clip_clip_loss + sent_sent_loss) of CycleConsistencyLoss.forward (coot/loss_fn.py:143-197)
    with num_samples = 1.  Returns the scalar loss (and the per-position losses if want_rows)."""
    if idx_clip is None:
        idx_clip = sample_cycle_indices(clip_lens)
    if idx_sent is None:
        idx_sent = sample_cycle_indices(sent_lens)
    B = clip_emb_reshape.shape[0] if global_batch is None else global_batch
    loss, rc, rs = _CycleConsFn.apply(clip_emb_reshape, sent_emb_reshape, clip_lens, sent_lens, idx_clip, idx_sent,
                                      weight, 1.0 / B, want_rows)
    return (loss, rc, rs) if want_rows else loss
