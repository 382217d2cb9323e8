"""
CPU baseline "port": the COOT retrieval train step restated with PyTorch CPU ops + autograd (fp32).

TEST / MEASUREMENT INFRASTRUCTURE ONLY — same rule as oracle/coot_oracle.py: nothing under coot-videotext_amd/
may import this.  It exists because the reference's own CPU path (PyTorch modules, `use_cuda=False`) cannot travel to
the GPU box (/root/reference is absent there), while BASELINE.json asks for "the reference's PyTorch CPU path timed
on the same box's host cores".  This module performs the same ATen operations in the same order as the reference
modules (nn.Linear -> addmm, softmax, erf-GELU, unbiased-std LayerNorm, autograd backward), so its wall time is a
faithful stand-in; its numerics are checked against the numpy oracle and the reference-generated golden fixtures
in tests/test_oracle_golden.py.

Each function cites the reference lines it restates (paths relative to the reference root).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

INF = 32752.0  # nntrainer/typext.py:24


def to_torch_params(P: Dict[str, np.ndarray], requires_grad: bool = True) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in P.items():
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).clone()
        if requires_grad and k != "embedding.pe":
            t.requires_grad_(True)
        out[k] = t
    return out


def ln_coot(x, gain, bias, eps=1e-6):
    """nntrainer/models/normalizations.py:98-101 (unbiased std, eps added to std)."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gain * (x - mean) / (std + eps) + bias


def mha(P, pre, xq, xkv, key_pad, H, p_drop=0.0, train=False):
    """MultiHeadAttention.forward (nntrainer/models/transformer_legacy.py:492-579)."""
    N, Lq, D = xq.shape
    Lk = xkv.shape[1]
    dh = D // H
    q = F.linear(xq, P[pre + "query_projection.weight"], P[pre + "query_projection.bias"])
    k = F.linear(xkv, P[pre + "key_projection.weight"], P[pre + "key_projection.bias"])
    v = F.linear(xkv, P[pre + "value_projection.weight"], P[pre + "value_projection.bias"])
    q = q.view(N, Lq, H, dh).transpose(1, 2)
    k = k.view(N, Lk, H, dh).transpose(1, 2)
    v = v.view(N, Lk, H, dh).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(dh)
    s = s.masked_fill(key_pad[:, None, None, :], -INF)
    a = F.dropout(torch.softmax(s, dim=-1), p_drop, train)
    ctx = torch.matmul(a, v).transpose(1, 2).contiguous().view(N, Lq, D)
    return F.linear(ctx, P[pre + "final_projection.weight"], P[pre + "final_projection.bias"])


def encoder_layer(P, pre, xq, xkv, key_pad, H, p_drop=0.0, train=False):
    """TransformerEncoderLayer.forward, post-LN sublayers (transformer_legacy.py:420-467, :582-605)."""
    a, f = pre + "self_attention_layer.", pre + "pointwise_feedforward_layer."
    att = F.dropout(mha(P, a + "sublayer.", xq, xkv, key_pad, H, p_drop, train), p_drop, train)
    x1 = ln_coot(att + xq, P[a + "layer_normalization.gain"], P[a + "layer_normalization.bias"])
    h = F.linear(x1, P[f + "sublayer.feed_forward.0.weight"], P[f + "sublayer.feed_forward.0.bias"])
    h = F.gelu(F.dropout(h, p_drop, train))
    h = F.dropout(F.linear(h, P[f + "sublayer.feed_forward.3.weight"], P[f + "sublayer.feed_forward.3.bias"]), p_drop, train)
    return ln_coot(h + x1, P[f + "layer_normalization.gain"], P[f + "layer_normalization.bias"])


def genpool(P, pre, x, pad, p_drop=0.0, train=False):
    """GenPool.forward (nntrainer/models/poolers.py:156-208)."""
    W1, b1, W2, b2 = (P[pre + n] for n in ("genpool_w1_head", "genpool_b1_head", "genpool_w2_head", "genpool_b2_head"))
    N, L, D = x.shape
    Hh = W1.shape[0]
    b1r = b1.unsqueeze(1).unsqueeze(0)
    b2r = b2.unsqueeze(1).unsqueeze(0)
    xe = x.unsqueeze(1)
    a = F.gelu(F.dropout(torch.matmul(xe, W1) + b1r, p_drop, train))
    s = F.dropout(torch.matmul(a, W2) + b2r, p_drop, train)
    s = s.masked_fill(pad[:, None, :, None], -INF)
    w = F.dropout(torch.softmax(s, dim=2), p_drop, train)
    w = w.transpose(1, 2).reshape(N, L, D)
    return (x * w).sum(1)


def net_fwd(P, cfg, feats, lengths, hidden=None, p_drop=0.0, train=False):
    """TransformerLegacy.forward (nntrainer/models/transformer_legacy.py:200-288)."""
    N, L, _ = feats.shape
    pad = torch.arange(L)[None, :] >= lengths[:, None]
    x = ln_coot(feats, P["norm_input.gain"], P["norm_input.bias"])
    if cfg.use_input_fc:
        x = F.gelu(F.linear(x, P["input_fc.mlp.0.weight"], P["input_fc.mlp.0.bias"]))
    x = x + P["embedding.pe"][:L]
    for i in range(cfg.num_layers):
        x = encoder_layer(P, f"tf.encoder_layers.{i}.", x, x, pad, cfg.num_heads, p_drop, train)
    ctx = None
    if cfg.use_context:
        cq = hidden.unsqueeze(1)
        for i in range(cfg.ctx_num_layers):
            cq = encoder_layer(P, f"tf_context.encoder_layers.{i}.", cq, x, pad, cfg.num_heads, p_drop, train)
        ctx = cq.squeeze(1)
    if cfg.pooler == "atn":
        pooled = genpool(P, "pooler.pools.0.", x, pad, p_drop, train)
    else:  # TemporalAvgPool "avg_special" (poolers.py:232-241): padded rows included in the sum
        pooled = x.sum(1) / lengths.unsqueeze(1).float()
    if ctx is not None:
        pooled = torch.cat([pooled, ctx], dim=-1)
    return pooled, x


def pack_by_count(emb, counts, cmax=None):
    """coot/model_retrieval.py:121-136 (the python loop, kept as a loop like the reference)."""
    B = len(counts)
    cmax = int(max(counts)) if cmax is None else cmax
    out = torch.zeros(B, cmax, emb.shape[1])
    mask = torch.ones(B, cmax, dtype=torch.bool)
    ptr = 0
    for b, c in enumerate(counts):
        c = int(c)
        out[b, :c] = emb[ptr:ptr + c]
        mask[b, :c] = False
        ptr += c
    return out, mask, torch.as_tensor(np.asarray(counts), dtype=torch.long)


def encode_side(Pl, cl, Pg, cg, ctx_feat, ctx_len, item_feat, item_len, item_num, p_drop=0.0, train=False):
    """encode_visual / encode_text (coot/model_retrieval.py:86-197)."""
    context, _ = net_fwd(Pl, cl, ctx_feat, ctx_len, None, p_drop, train)
    item_emb, _ = net_fwd(Pl, cl, item_feat, item_len, None, p_drop, train)
    resh, mask, lens = pack_by_count(item_emb, item_num)
    glob, _ = net_fwd(Pg, cg, resh, lens, context, p_drop, train)
    return dict(global_emb=glob, item_emb=item_emb, context=context, item_emb_reshape=resh, item_emb_mask=mask, item_emb_lens=lens)


def contrastive(im, s, margin):
    """ContrastiveLoss.forward (coot/loss_fn.py:63-100), max_violation False."""
    scores = im @ s.t()
    diag = scores.diag().view(-1, 1)
    eye = torch.eye(scores.shape[0], dtype=torch.bool, device=scores.device)
    cost_s = (margin + scores - diag).clamp(min=0).masked_fill(eye, 0)
    cost_im = (margin + scores - diag.t()).clamp(min=0).masked_fill(eye, 0)
    return (cost_s.sum() + cost_im.sum()) / (scores.shape[0] ** 2)


def total_contrastive(vis, txt, w, margin):
    """compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182), incl. the :181 weight quirk."""
    n = lambda t: F.normalize(t)  # noqa: E731
    ve, pe, ce, se = n(vis["global_emb"]), n(txt["global_emb"]), n(vis["item_emb"]), n(txt["item_emb"])
    vc, pc = n(vis["context"]), n(txt["context"])
    loss = 0
    for wt, a, b in ((w["weight_high"], ve, pe), (w["weight_low"], ce, se), (w["weight_context"], vc, pc)):
        if wt != 0:
            loss = loss + wt * contrastive(a, b, margin)
    for wt, a, b in ((w["weight_high_internal"], ve, pe), (w["weight_low_internal"], ce, se),
                     (w["weight_low_internal"] if w["weight_context_internal"] != 0 else 0, vc, pc)):
        if wt != 0:
            loss = loss + wt * (contrastive(a, a, margin) + contrastive(b, b, margin)) / 2
    return loss


def _soft_nn(src, src_pad, tgt, tgt_pad):
    """CycleConsistencyLoss.get_soft_nn (coot/loss_fn.py:226-268)."""
    d = -((src.unsqueeze(2) - tgt.unsqueeze(1)) ** 2).mean(-1)
    tot = src_pad.unsqueeze(2) | tgt_pad.unsqueeze(1)
    w = torch.softmax(d.masked_fill(tot, -INF), dim=-1)
    return (tgt.unsqueeze(1) * w.unsqueeze(-1)).sum(2), w


def cyclecons(clip, clip_pad, sent, sent_pad, idx_clip, idx_sent):
    """CycleConsistencyLoss.forward + get_total_loss(num_samples=1) (coot/loss_fn.py:143-319), index-simple loss."""
    def rows(a, ap, b, bp):
        nn1, _ = _soft_nn(a, ap, b, bp)
        _, beta = _soft_nn(nn1, ap, a, ap)
        ar = torch.arange(a.shape[1], device=a.device).float()
        mu = (beta * ar[None, None, :]).sum(-1)
        return ((mu - ar[None, :]) ** 2).masked_fill(ap, 0)
    B = clip.shape[0]
    lc, ls = rows(clip, clip_pad, sent, sent_pad), rows(sent, sent_pad, clip, clip_pad)
    ar = torch.arange(B, device=clip.device)
    return lc[ar, idx_clip].mean(), ls[ar, idx_sent].mean()


def full_step(cfgs, Ps: List[Dict[str, torch.Tensor]], b, idx_clip, idx_sent, w, margin=0.2, cc_weight=0.01, p_drop=0.0,
              train=False, backward=True):
    """encode_visual + encode_text + losses (+ backward): the timed body of coot/trainer_retrieval.py:253-291.
    `b` is the numpy batch dict of oracle.coot_oracle.make_batch.  Returns (vis, txt, contr, cc)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    li = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64))  # noqa: E731
    vis = encode_side(Ps[0], cfgs[0], Ps[1], cfgs[1], t(b["vid_feat"]), li(b["vid_feat_len"]), t(b["clip_feat"]), li(b["clip_feat_len"]),
                      b["clip_num"], p_drop, train)
    txt = encode_side(Ps[2], cfgs[2], Ps[3], cfgs[3], t(b["par_feat"]), li(b["par_feat_len"]), t(b["sent_feat"]), li(b["sent_feat_len"]),
                      b["sent_num"], p_drop, train)
    contr = total_contrastive(vis, txt, w, margin)
    cc = torch.zeros(())
    if cc_weight != 0:
        c1, c2 = cyclecons(vis["item_emb_reshape"], vis["item_emb_mask"], txt["item_emb_reshape"], txt["item_emb_mask"], li(idx_clip), li(idx_sent))
        cc = cc_weight * (c1 + c2)
    if backward:
        for P in Ps:
            for v in P.values():
                if v.grad is not None:
                    v.grad = None
        (contr + cc).backward()
    return vis, txt, contr, cc
