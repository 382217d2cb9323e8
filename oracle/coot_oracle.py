"""
CPU oracle for the COOT retrieval hot path (numpy restatement of the reference algorithm).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (coot-videotext_amd/) may import
this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
checker.  The product path fails loudly when the HIP library is missing.

Parity pinning: the reference holds NO numeric golden vectors for this path (tests_coot/ is
empty).  This oracle is therefore pinned against outputs of the reference
itself, generated in the build container by oracle/gen_golden.py (imports /root/reference, runs
its unmodified modules on seeded inputs) and committed under tests/golden/*.npz.
tests/test_oracle_golden.py checks every function below against those fixtures.

Every function cites the reference file:line it restates (paths relative to the reference root).

Numerics: default dtype float64 (the reference's CPU path is float32; fixtures agree to ~1e-6).
`Rounding` lets the oracle emulate the bf16 rounding points of the HIP dataflow (operands of every
MFMA and every bf16 tensor written to HBM) so that HIP-vs-oracle parity can be checked tightly
and bf16-vs-fp32 drift can be measured separately.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

try:  # scipy is in the image; keep a pure-numpy fallback so the oracle never silently degrades
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

INF = 32752.0  # nntrainer/typext.py:24  (mask fill value, NOT -inf)
LN_EPS = 1e-6  # nntrainer/models/normalizations.py:89


# ---------------------------------------------------------------------------------------------
# bf16 emulation
# ---------------------------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 -> float (same as v_cvt_pk_bf16_f32)."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    u = (u + 0x7FFF + lsb) & 0xFFFF0000
    out = u.astype(np.uint32).view(np.float32)
    # NaN stays NaN (not needed on this path, inputs are finite)
    return out.astype(x.dtype if x.dtype in (np.float32, np.float64) else np.float32)


class Rounding:
    """q(x): identity (exact mode) or bf16 rounding (emulation of the HIP dataflow)."""

    def __init__(self, bf16: bool = False):
        self.bf16 = bf16

    def __call__(self, x: np.ndarray) -> np.ndarray:
        return bf16_round(x) if self.bf16 else x


EXACT = Rounding(False)
BF16 = Rounding(True)


# ---------------------------------------------------------------------------------------------
# configuration (mirrors the fields of nntrainer/models/transformer_legacy.py:26-97 that the
# shipped YAMLs use; everything else is asserted off by the host config loader)
# ---------------------------------------------------------------------------------------------
@dataclass
class NetConfig:
    input_dim: int
    hidden_dim: int = 384
    num_heads: int = 8
    ff_dim: int = 384  # pointwise_ff_dim (0 => hidden_dim, transformer_legacy.py:409-410)
    num_layers: int = 1
    use_input_fc: bool = True
    use_context: bool = False
    ctx_num_layers: int = 1
    pooler: str = "atn"  # "atn" | "avg_special"
    pool_hidden: int = 768  # GenPool d_attn (0 => hidden_dim, poolers.py:121-122)
    pool_heads: int = 2

    def __post_init__(self):
        if self.ff_dim == 0:
            self.ff_dim = self.hidden_dim
        if self.pool_hidden == 0:
            self.pool_hidden = self.hidden_dim


# ---------------------------------------------------------------------------------------------
# elementary ops
# ---------------------------------------------------------------------------------------------
def gelu(x):
    """nn.GELU() exact-erf form (nntrainer/models/activations.py:29-30)."""
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    """d/dx GELU = Phi(x) + x*phi(x)  (SURVEY appendix A.7)."""
    return 0.5 * (1.0 + _erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def ln_coot(x, gain, bias, eps=LN_EPS):
    """LayerNormalization 'layernorm_coot' (nntrainer/models/normalizations.py:98-101):
    gain * (x - mean) / (std_unbiased + eps) + bias over the last dim."""
    n = x.shape[-1]
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    std = np.sqrt((xc * xc).sum(-1, keepdims=True) / (n - 1))
    return gain * xc / (std + eps) + bias


def ln_coot_xhat(x, eps=LN_EPS):
    n = x.shape[-1]
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    std = np.sqrt((xc * xc).sum(-1, keepdims=True) / (n - 1))
    return xc / (std + eps)


def ln_coot_bwd(dy, x, gain, eps=LN_EPS):
    """Backward of ln_coot (verified against autograd by the golden test).
    Returns dx, dgain, dbias.  Rows with std == 0 (zero padding) get only the first term, as
    PyTorch's std backward yields 0 there."""
    n = x.shape[-1]
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    std = np.sqrt((xc * xc).sum(-1, keepdims=True) / (n - 1))
    s = std + eps
    h = dy * gain
    t1 = (h - h.mean(-1, keepdims=True)) / s
    dot = (h * xc).sum(-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t2 = np.where(std > 0, dot / (s * s) * xc / ((n - 1) * np.where(std > 0, std, 1.0)), 0.0)
    dx = t1 - t2
    red = tuple(range(dy.ndim - 1))
    dgain = (dy * xc / s).sum(red)
    dbias = dy.sum(red)
    return dx, dgain, dbias


def sincos_pe(max_len: int, dim: int) -> np.ndarray:
    """PositionalEncodingSinCos buffer (nntrainer/models/encoder.py:84-90).  Note the
    non-standard exponent: 10000 ** (2 * d / dim) uses the dim index d itself.
    Computed in float32 like the reference (the buffer is part of the state dict)."""
    pe = np.zeros((max_len, dim), dtype=np.float32)
    position = np.arange(max_len, dtype=np.float32)[:, None]
    dimension = np.arange(dim, dtype=np.float32)
    div_term = np.float32(10000.0) ** (np.float32(2.0) * dimension / np.float32(dim))
    pe[:, 0::2] = np.sin(position / div_term[0::2])
    pe[:, 1::2] = np.cos(position / div_term[1::2])
    return pe


def masked_softmax_lastdim(s):
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(-1, keepdims=True)


# ---------------------------------------------------------------------------------------------
# multi-head attention + encoder layer (post-LN)
# ---------------------------------------------------------------------------------------------
def _p(P, prefix, name):
    return P[prefix + name]


def mha_fwd(P, pre, xq, xkv, key_valid, num_heads, q: Rounding = EXACT):
    """MultiHeadAttention.forward (nntrainer/models/transformer_legacy.py:492-579), eval mode.
    xq [N,Lq,D], xkv [N,Lk,D], key_valid [N,Lk] bool (True = real key).  All Lq query rows are
    computed; only keys are masked, with -INF fill (:544-545)."""
    N, Lq, D = xq.shape
    Lk = xkv.shape[1]
    H = num_heads
    dh = D // H
    Wq, bq = q(_p(P, pre, "query_projection.weight")), _p(P, pre, "query_projection.bias")
    Wk, bk = q(_p(P, pre, "key_projection.weight")), _p(P, pre, "key_projection.bias")
    Wv, bv = q(_p(P, pre, "value_projection.weight")), _p(P, pre, "value_projection.bias")
    Wo, bo = q(_p(P, pre, "final_projection.weight")), _p(P, pre, "final_projection.bias")
    Q = q(xq @ Wq.T + bq)
    K = q(xkv @ Wk.T + bk)
    V = q(xkv @ Wv.T + bv)
    Qh = Q.reshape(N, Lq, H, dh).transpose(0, 2, 1, 3)
    Kh = K.reshape(N, Lk, H, dh).transpose(0, 2, 1, 3)
    Vh = V.reshape(N, Lk, H, dh).transpose(0, 2, 1, 3)
    S = Qh @ Kh.transpose(0, 1, 3, 2) / math.sqrt(dh)
    S = np.where(key_valid[:, None, None, :], S, -INF)
    A = masked_softmax_lastdim(S)
    ctx = (q(A) @ Vh).transpose(0, 2, 1, 3).reshape(N, Lq, D)
    ctx = q(ctx)
    out = ctx @ Wo.T + bo
    cache = dict(xq=xq, xkv=xkv, Qh=Qh, Kh=Kh, Vh=Vh, A=A, ctx=ctx, key_valid=key_valid)
    return out, cache


def mha_bwd(P, pre, dout, cache, num_heads):
    """Backward of mha_fwd.  Returns dxq, dxkv (separately; caller sums for self-attention)
    and parameter grads."""
    xq, xkv, Qh, Kh, Vh, A, ctx = (cache[k] for k in ("xq", "xkv", "Qh", "Kh", "Vh", "A", "ctx"))
    N, Lq, D = xq.shape
    Lk = xkv.shape[1]
    H = num_heads
    dh = D // H
    Wq = _p(P, pre, "query_projection.weight")
    Wk = _p(P, pre, "key_projection.weight")
    Wv = _p(P, pre, "value_projection.weight")
    Wo = _p(P, pre, "final_projection.weight")
    G = {}
    d2 = dout.reshape(-1, D)
    G[pre + "final_projection.weight"] = d2.T @ ctx.reshape(-1, D)
    G[pre + "final_projection.bias"] = d2.sum(0)
    dctx = dout @ Wo
    dctxh = dctx.reshape(N, Lq, H, dh).transpose(0, 2, 1, 3)
    dA = dctxh @ Vh.transpose(0, 1, 3, 2)
    dVh = A.transpose(0, 1, 3, 2) @ dctxh
    dS = A * (dA - (dA * A).sum(-1, keepdims=True))
    # masked entries have A == 0 exactly -> dS == 0 there (masked_fill blocks the grad)
    dS = dS / math.sqrt(dh)
    dQh = dS @ Kh
    dKh = dS.transpose(0, 1, 3, 2) @ Qh
    dQ = dQh.transpose(0, 2, 1, 3).reshape(N, Lq, D)
    dK = dKh.transpose(0, 2, 1, 3).reshape(N, Lk, D)
    dV = dVh.transpose(0, 2, 1, 3).reshape(N, Lk, D)
    xq2, xkv2 = xq.reshape(-1, D), xkv.reshape(-1, D)
    G[pre + "query_projection.weight"] = dQ.reshape(-1, D).T @ xq2
    G[pre + "query_projection.bias"] = dQ.reshape(-1, D).sum(0)
    G[pre + "key_projection.weight"] = dK.reshape(-1, D).T @ xkv2
    G[pre + "key_projection.bias"] = dK.reshape(-1, D).sum(0)
    G[pre + "value_projection.weight"] = dV.reshape(-1, D).T @ xkv2
    G[pre + "value_projection.bias"] = dV.reshape(-1, D).sum(0)
    dxq = dQ @ Wq
    dxkv = dK @ Wk + dV @ Wv
    return dxq, dxkv, G


def encoder_layer_fwd(P, pre, xq, xkv, key_valid, num_heads, q: Rounding = EXACT):
    """TransformerEncoderLayer.forward (transformer_legacy.py:420-438) with Sublayer post-LN
    (:453-464), eval mode (all dropouts identity):
        x1 = LN(MHA(xq, xkv, xkv) + xq);  out = LN(FF(x1) + x1)
    FF = Linear -> [Dropout] -> GELU -> Linear -> [Dropout]  (:592-598)."""
    a = pre + "self_attention_layer."
    f = pre + "pointwise_feedforward_layer."
    att, c_att = mha_fwd(P, a + "sublayer.", xq, xkv, key_valid, num_heads, q)
    r1 = q(att + xq)
    x1 = q(ln_coot(r1, P[a + "layer_normalization.gain"], P[a + "layer_normalization.bias"]))
    W1, b1 = q(P[f + "sublayer.feed_forward.0.weight"]), P[f + "sublayer.feed_forward.0.bias"]
    W2, b2 = q(P[f + "sublayer.feed_forward.3.weight"]), P[f + "sublayer.feed_forward.3.bias"]
    h1 = x1 @ W1.T + b1
    a1 = q(gelu(h1))
    r2 = q(a1 @ W2.T + b2 + x1)
    out = q(ln_coot(r2, P[f + "layer_normalization.gain"], P[f + "layer_normalization.bias"]))
    cache = dict(c_att=c_att, r1=r1, x1=x1, h1=h1, a1=a1, r2=r2, self_attn=(xq is xkv))
    return out, cache


def encoder_layer_bwd(P, pre, dout, cache, num_heads):
    a = pre + "self_attention_layer."
    f = pre + "pointwise_feedforward_layer."
    G = {}
    D = dout.shape[-1]
    dr2, dg, db = ln_coot_bwd(dout, cache["r2"], P[f + "layer_normalization.gain"])
    G[f + "layer_normalization.gain"], G[f + "layer_normalization.bias"] = dg, db
    W1 = P[f + "sublayer.feed_forward.0.weight"]
    W2 = P[f + "sublayer.feed_forward.3.weight"]
    d2 = dr2.reshape(-1, D)
    G[f + "sublayer.feed_forward.3.weight"] = d2.T @ cache["a1"].reshape(-1, cache["a1"].shape[-1])
    G[f + "sublayer.feed_forward.3.bias"] = d2.sum(0)
    da1 = dr2 @ W2
    dh1 = da1 * gelu_grad(cache["h1"])
    F = dh1.shape[-1]
    G[f + "sublayer.feed_forward.0.weight"] = dh1.reshape(-1, F).T @ cache["x1"].reshape(-1, D)
    G[f + "sublayer.feed_forward.0.bias"] = dh1.reshape(-1, F).sum(0)
    dx1 = dr2 + dh1 @ W1
    dr1, dg, db = ln_coot_bwd(dx1, cache["r1"], P[a + "layer_normalization.gain"])
    G[a + "layer_normalization.gain"], G[a + "layer_normalization.bias"] = dg, db
    dxq, dxkv, Ga = mha_bwd(P, a + "sublayer.", dr1, cache["c_att"], num_heads)
    G.update(Ga)
    dxq = dxq + dr1  # residual
    return dxq, dxkv, G


# ---------------------------------------------------------------------------------------------
# poolers
# ---------------------------------------------------------------------------------------------
def genpool_fwd(P, pre, x, valid, q: Rounding = EXACT):
    """GenPool.forward (nntrainer/models/poolers.py:156-208), eval mode.  x [N,L,D],
    valid [N,L] bool.  Per head h: a = GELU(x W1[h] + b1[h]); s = a W2[h] + b2[h];
    padded rows <- -INF; softmax over the sequence axis per channel; heads concatenated on the
    channel axis (transpose(1,2).reshape, :197-199); pooled = sum_l x * w."""
    W1, b1 = P[pre + "genpool_w1_head"], P[pre + "genpool_b1_head"]  # [H,D,dh], [H,dh]
    W2, b2 = P[pre + "genpool_w2_head"], P[pre + "genpool_b2_head"]  # [H,dh,do], [H,do]
    H = W1.shape[0]
    hp = np.einsum("nld,hde->nhle", x, q(W1)) + b1[None, :, None, :]
    ap = q(gelu(hp))
    s = q(np.einsum("nhle,heo->nhlo", ap, q(W2)) + b2[None, :, None, :])
    s = np.where(valid[:, None, :, None], s, -INF)
    m = s.max(2, keepdims=True)
    e = np.exp(s - m)
    w = e / e.sum(2, keepdims=True)  # softmax over L
    N, _, L, do = w.shape
    wcat = w.transpose(0, 2, 1, 3).reshape(N, L, H * do)
    pooled = (x * wcat).sum(1)
    cache = dict(x=x, hp=hp, ap=ap, wcat=wcat, pooled=pooled, valid=valid, H=H)
    return pooled, cache


def genpool_bwd(P, pre, dpooled, cache):
    """SURVEY appendix A.7."""
    x, hp, ap, wcat, pooled, H = (cache[k] for k in ("x", "hp", "ap", "wcat", "pooled", "H"))
    W1, W2 = P[pre + "genpool_w1_head"], P[pre + "genpool_w2_head"]
    N, L, D = x.shape
    do = W2.shape[2]
    G = {}
    dw = dpooled[:, None, :] * x
    ds = wcat * (dw - (wcat * dw).sum(1, keepdims=True))  # [N,L,H*do]; padded rows: w==0 -> 0
    dsh = ds.reshape(N, L, H, do).transpose(0, 2, 1, 3)  # [N,H,L,do]
    G[pre + "genpool_w2_head"] = np.einsum("nhle,nhlo->heo", ap, dsh)
    G[pre + "genpool_b2_head"] = dsh.sum((0, 2))
    dap = np.einsum("nhlo,heo->nhle", dsh, W2)
    dhp = dap * gelu_grad(hp)
    G[pre + "genpool_w1_head"] = np.einsum("nld,nhle->hde", x, dhp)
    G[pre + "genpool_b1_head"] = dhp.sum((0, 2))
    dx = dpooled[:, None, :] * wcat + np.einsum("nhle,hde->nld", dhp, W1)
    return dx, G


# ---------------------------------------------------------------------------------------------
# TransformerLegacy (one COOT network)
# ---------------------------------------------------------------------------------------------
def net_fwd(P, cfg: NetConfig, feats, lengths, hidden_state=None, q: Rounding = EXACT):
    """TransformerLegacy.forward (nntrainer/models/transformer_legacy.py:200-288), eval mode,
    for the options the shipped configs use.
    feats [N,L,Din] zero padded, lengths [N] int, hidden_state [N,D] or None.
    Returns (pooled, per_token, cache)."""
    N, L, Din = feats.shape
    D, H = cfg.hidden_dim, cfg.num_heads
    valid = np.arange(L)[None, :] < np.asarray(lengths)[:, None]
    g0, b0 = P["norm_input.gain"], P["norm_input.bias"]
    pe = P["embedding.pe"][:L].astype(feats.dtype)
    cache = dict(feats=feats, valid=valid, lengths=np.asarray(lengths))
    if cfg.use_input_fc:
        # HIP dataflow: xhat is the bf16 operand, LN affine folded into the FC weight/bias
        Win, bin_ = P["input_fc.mlp.0.weight"], P["input_fc.mlp.0.bias"]
        if q.bf16:
            xhat = q(ln_coot_xhat(feats))
            h0 = xhat @ q(Win * g0[None, :]).T + (bin_ + Win @ b0)
        else:
            h0 = ln_coot(feats, g0, b0) @ Win.T + bin_
        z = q(gelu(h0) + pe)
        cache["h0"] = h0
    else:
        z = q(ln_coot(feats, g0, b0) + pe)
    cache["z0"] = z
    layer_caches = []
    for i in range(cfg.num_layers):
        z, c = encoder_layer_fwd(P, f"tf.encoder_layers.{i}.", z, z, valid, H, q)
        layer_caches.append(c)
    cache["layers"] = layer_caches
    cache["zL"] = z
    ctx = None
    if cfg.use_context:
        assert hidden_state is not None
        cq = hidden_state[:, None, :]
        ctx_caches = []
        for i in range(cfg.ctx_num_layers):
            cq, c = encoder_layer_fwd(P, f"tf_context.encoder_layers.{i}.", cq, z, valid, H, q)
            ctx_caches.append(c)
        cache["ctx_layers"] = ctx_caches
        ctx = cq[:, 0, :]
    if cfg.pooler == "atn":
        pooled, pc = genpool_fwd(P, "pooler.pools.0.", z, valid, q)
        cache["pool"] = pc
    elif cfg.pooler == "avg_special":
        # TemporalAvgPool (poolers.py:232-241): sums ALL rows incl. padding, divides by length
        pooled = z.sum(1) / np.asarray(lengths, dtype=z.dtype)[:, None]
    else:
        raise NotImplementedError(cfg.pooler)
    if ctx is not None:
        pooled = np.concatenate([pooled, ctx], -1)
    return pooled, z, cache


def net_bwd(P, cfg: NetConfig, dpooled, cache, need_dfeats=False):
    """Backward of net_fwd wrt parameters (+ hidden_state, + feats for the global nets)."""
    D, H = cfg.hidden_dim, cfg.num_heads
    feats, valid, lengths = cache["feats"], cache["valid"], cache["lengths"]
    N, L, Din = feats.shape
    G: Dict[str, np.ndarray] = {}
    z = cache["zL"]
    dhidden = None
    if cfg.use_context:
        dpool, dctx = dpooled[:, :-D], dpooled[:, -D:]
    else:
        dpool, dctx = dpooled, None
    if cfg.pooler == "atn":
        dz, Gp = genpool_bwd(P, "pooler.pools.0.", dpool, cache["pool"])
        G.update(Gp)
    else:
        dz = np.broadcast_to((dpool / lengths[:, None].astype(dpool.dtype))[:, None, :], z.shape).copy()
    if cfg.use_context:
        dcq = dctx[:, None, :]
        for i in reversed(range(cfg.ctx_num_layers)):
            dcq, dkv, Gc = encoder_layer_bwd(P, f"tf_context.encoder_layers.{i}.", dcq,
                                             cache["ctx_layers"][i], H)
            for k, v in Gc.items():
                G[k] = G.get(k, 0) + v
            dz = dz + dkv
        dhidden = dcq[:, 0, :]
    for i in reversed(range(cfg.num_layers)):
        dxq, dxkv, Gl = encoder_layer_bwd(P, f"tf.encoder_layers.{i}.", dz, cache["layers"][i], H)
        G.update(Gl)
        dz = dxq + dxkv
    g0, b0 = P["norm_input.gain"], P["norm_input.bias"]
    dfeats = None
    if cfg.use_input_fc:
        Win = P["input_fc.mlp.0.weight"]
        dh0 = dz * gelu_grad(cache["h0"])
        u = ln_coot(feats, g0, b0)
        G["input_fc.mlp.0.weight"] = dh0.reshape(-1, D).T @ u.reshape(-1, Din)
        G["input_fc.mlp.0.bias"] = dh0.reshape(-1, D).sum(0)
        du = dh0 @ Win
        dfeats, dg, db = ln_coot_bwd(du, feats, g0)
    else:
        dfeats, dg, db = ln_coot_bwd(dz, feats, g0)
    G["norm_input.gain"], G["norm_input.bias"] = dg, db
    return G, dhidden, (dfeats if need_dfeats else None)


# ---------------------------------------------------------------------------------------------
# encode_visual / encode_text
# ---------------------------------------------------------------------------------------------
def pack_by_count(emb, counts, cmax=None):
    """The python pack loop of RetrievalModelManager.encode_visual (coot/model_retrieval.py:121-136):
    flat [Nc,D] -> zero padded [B,Cmax,D], mask (True = pad), lens."""
    counts = np.asarray(counts)
    B = len(counts)
    cmax = int(counts.max()) if cmax is None else cmax
    out = np.zeros((B, cmax, emb.shape[1]), dtype=emb.dtype)
    mask = np.ones((B, cmax), dtype=bool)
    ptr = 0
    for b, c in enumerate(counts):
        out[b, :c] = emb[ptr:ptr + c]
        mask[b, :c] = False
        ptr += c
    return out, mask, counts.copy()


def unpack_by_count(demb_reshape, counts):
    parts = [demb_reshape[b, :c] for b, c in enumerate(counts)]
    return np.concatenate(parts, 0)


def encode_side(P_local, cfg_local, P_global, cfg_global, ctx_feat, ctx_len, item_feat, item_len,
                item_num, q: Rounding = EXACT, cmax=None):
    """encode_visual / encode_text (coot/model_retrieval.py:86-141 / :143-197):
    context = Local(vid/par feats); item_emb = Local(clip/sent feats) [same weights];
    pack by item_num; global_emb = Global(packed, context).
    `cmax` overrides max(item_num) (data-parallel: global max over ranks)."""
    context, _, c_ctx = net_fwd(P_local, cfg_local, ctx_feat, ctx_len, None, q)
    item_emb, _, c_item = net_fwd(P_local, cfg_local, item_feat, item_len, None, q)
    reshaped, mask, lens = pack_by_count(item_emb, item_num, cmax)
    glob, _, c_glob = net_fwd(P_global, cfg_global, reshaped, lens, context, q)
    out = dict(global_emb=glob, item_emb=item_emb, context=context, item_emb_reshape=reshaped,
               item_emb_mask=mask, item_emb_lens=lens)
    cache = dict(c_ctx=c_ctx, c_item=c_item, c_glob=c_glob, item_num=np.asarray(item_num))
    return out, cache


def encode_side_bwd(P_local, cfg_local, P_global, cfg_global, cache, d_global, d_item, d_context,
                    d_item_reshape=None):
    """Backward of encode_side.  d_item_reshape = grad flowing into the packed tensor directly
    (cycle-consistency loss reads clip_emb_reshape)."""
    Gg, dhidden, dresh = net_bwd(P_global, cfg_global, d_global, cache["c_glob"], need_dfeats=True)
    if d_item_reshape is not None:
        dresh = dresh + d_item_reshape
    d_item_total = d_item + unpack_by_count(dresh, cache["item_num"])
    d_ctx_total = d_context + dhidden
    Gl1, _, _ = net_bwd(P_local, cfg_local, d_item_total, cache["c_item"])
    Gl2, _, _ = net_bwd(P_local, cfg_local, d_ctx_total, cache["c_ctx"])
    Gl = {k: Gl1[k] + Gl2[k] for k in Gl1}
    return Gl, Gg


# ---------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------
def l2_normalize(v, eps=1e-12):
    """F.normalize(p=2, dim=1, eps=1e-12) (coot/trainer_retrieval.py:161-166)."""
    n = np.sqrt((v * v).sum(1, keepdims=True))
    return v / np.maximum(n, eps)


def l2_normalize_bwd(da, v, eps=1e-12):
    n = np.sqrt((v * v).sum(1, keepdims=True))
    nn_ = np.maximum(n, eps)
    a = v / nn_
    return (da - a * (a * da).sum(1, keepdims=True)) / nn_


def contrastive_loss(im, s, margin, q: Rounding = EXACT):
    """ContrastiveLoss.forward, max_violation=False, norm=True (coot/loss_fn.py:63-100).
    Returns loss, d_im, d_s  (SURVEY appendix A.4)."""
    N = im.shape[0]
    S = q(im) @ q(s).T
    diag = np.diag(S)
    cs = margin + S - diag[:, None]
    ci = margin + S - diag[None, :]
    off = ~np.eye(N, dtype=bool)
    ms = (cs > 0) & off
    mi = (ci > 0) & off
    loss = (cs[ms].sum() + ci[mi].sum()) / (N * N)
    G = (ms.astype(S.dtype) + mi.astype(S.dtype))
    gd = -(ms.sum(1) + mi.sum(0)).astype(S.dtype)
    G[np.arange(N), np.arange(N)] = gd
    G /= (N * N)
    return loss, G @ s, G.T @ im


def total_contrastive_loss(E: Dict[str, np.ndarray], w: Dict[str, float], margin: float,
                           q: Rounding = EXACT):
    """compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182), including the
    reference quirk that the context-internal term is weighted by weight_low_internal (:181).
    E keys: vid_emb, par_emb, clip_emb, sent_emb, vid_context, par_context (un-normalised).
    Returns loss and grads wrt the un-normalised embeddings."""
    keys = ["vid_emb", "par_emb", "clip_emb", "sent_emb", "vid_context", "par_context"]
    A = {k: l2_normalize(E[k]) for k in keys}
    dA = {k: np.zeros_like(E[k]) for k in keys}
    loss = 0.0

    def align(kv, kt, wt):
        nonlocal loss
        l, dv, dt = contrastive_loss(A[kv], A[kt], margin, q)
        loss += wt * l
        dA[kv] += wt * dv
        dA[kt] += wt * dt

    def cluster(kv, kt, wt):
        nonlocal loss
        for k in (kv, kt):
            l, d1, d2 = contrastive_loss(A[k], A[k], margin, q)
            loss += wt * 0.5 * l
            dA[k] += wt * 0.5 * (d1 + d2)

    if w["weight_high"] != 0:
        align("vid_emb", "par_emb", w["weight_high"])
    if w["weight_low"] != 0:
        align("clip_emb", "sent_emb", w["weight_low"])
    if w["weight_context"] != 0:
        align("vid_context", "par_context", w["weight_context"])
    if w["weight_high_internal"] != 0:
        cluster("vid_emb", "par_emb", w["weight_high_internal"])
    if w["weight_low_internal"] != 0:
        cluster("clip_emb", "sent_emb", w["weight_low_internal"])
    if w["weight_context_internal"] != 0:
        cluster("vid_context", "par_context", w["weight_low_internal"])  # sic, :181
    dE = {k: l2_normalize_bwd(dA[k], E[k]) for k in keys}
    return loss, dE


def _soft_nn(src, src_valid, tgt, tgt_valid):
    """CycleConsistencyLoss.get_soft_nn (coot/loss_fn.py:226-268): negative mean-squared
    distance, masked with -INF where source OR target is padding, softmax over targets."""
    d = -((src[:, :, None, :] - tgt[:, None, :, :]) ** 2).mean(-1)
    tot = src_valid[:, :, None] & tgt_valid[:, None, :]
    d = np.where(tot, d, -INF)
    w = masked_softmax_lastdim(d)
    nn_ = (tgt[:, None, :, :] * w[:, :, :, None]).sum(2)
    return nn_, w


def cycle_consistency_rows(clip, clip_valid, sent, sent_valid):
    """Per-position cycle loss l[b,i] = (mu_i - i)^2 of one direction
    (coot/loss_fn.py:143-197, :321-387 with weight_index_simple=1, weight_index_gauss=0).
    Rows with clip_valid False are 0 (masked, :361)."""
    nn1, _ = _soft_nn(clip, clip_valid, sent, sent_valid)
    _, beta = _soft_nn(nn1, clip_valid, clip, clip_valid)
    C = clip.shape[1]
    mu = (beta * np.arange(C)[None, None, :]).sum(-1)
    l = (mu - np.arange(C)[None, :]) ** 2
    return np.where(clip_valid, l, 0.0)


def cycle_consistency_loss(clip, clip_valid, sent, sent_valid, idx_clip, idx_sent):
    """CycleConsistencyLoss.forward + get_total_loss with num_samples=1 (coot/loss_fn.py:270-319):
    loss = mean_b l[b, idx[b]] for each direction, idx drawn by th.multinomial over valid
    positions (supplied here).  Returns (clip_clip_loss, sent_sent_loss)."""
    B = clip.shape[0]
    lc = cycle_consistency_rows(clip, clip_valid, sent, sent_valid)
    ls = cycle_consistency_rows(sent, sent_valid, clip, clip_valid)
    return lc[np.arange(B), idx_clip].mean(), ls[np.arange(B), idx_sent].mean()


def _cc_dir_bwd(src, src_valid, tgt, tgt_valid, idx, scale):
    """Gradient of scale * mean_b l[b, idx[b]] (one direction) wrt src and tgt.  Only the sampled
    row i* of each video carries gradient."""
    B, Cs, D = src.shape
    Ct = tgt.shape[1]
    dsrc = np.zeros_like(src)
    dtgt = np.zeros_like(tgt)
    for b in range(B):
        i = int(idx[b])
        tv = tgt_valid[b]
        sv = src_valid[b]
        c = src[b, i]
        dist = np.where(tv, -((c[None, :] - tgt[b]) ** 2).mean(-1), -INF)
        alpha = masked_softmax_lastdim(dist[None, :])[0]
        nn1 = (alpha[:, None] * tgt[b]).sum(0)
        dist2 = np.where(sv, -((nn1[None, :] - src[b]) ** 2).mean(-1), -INF)
        beta = masked_softmax_lastdim(dist2[None, :])[0]
        ks = np.arange(Cs)
        mu = (beta * ks).sum()
        dmu = scale / B * 2.0 * (mu - i)
        dbeta = dmu * ks
        ddist2 = beta * (dbeta - (beta * dbeta).sum())
        ddist2 = np.where(sv, ddist2, 0.0)
        diff2 = nn1[None, :] - src[b]  # [Cs,D]
        dnn1 = (ddist2[:, None] * (-2.0 / D) * diff2).sum(0)
        dsrc[b] += ddist2[:, None] * (2.0 / D) * diff2
        dalpha = tgt[b] @ dnn1
        dtgt[b] += alpha[:, None] * dnn1[None, :]
        ddist = alpha * (dalpha - (alpha * dalpha).sum())
        ddist = np.where(tv, ddist, 0.0)
        diff = c[None, :] - tgt[b]  # [Ct,D]
        dsrc[b, i] += (ddist[:, None] * (-2.0 / D) * diff).sum(0)
        dtgt[b] += ddist[:, None] * (2.0 / D) * diff
    return dsrc, dtgt


def cycle_consistency_bwd(clip, clip_valid, sent, sent_valid, idx_clip, idx_sent, weight):
    """Gradient of weight * (clip_clip_loss + sent_sent_loss) wrt clip and sent."""
    dc1, ds1 = _cc_dir_bwd(clip, clip_valid, sent, sent_valid, idx_clip, weight)
    ds2, dc2 = _cc_dir_bwd(sent, sent_valid, clip, clip_valid, idx_sent, weight)
    return dc1 + dc2, ds1 + ds2


# ---------------------------------------------------------------------------------------------
# optimizers (nntrainer/optimization.py): Adam as torch.optim.Adam (coupled L2), RAdam as the in-file class (:79-181)
# ---------------------------------------------------------------------------------------------
def radam_scalars(step: int, beta1: float, beta2: float, degenerated_to_sgd: bool):
    """nntrainer/optimization.py:144-164: (mode, step_size).  mode 'rect': adaptive update with the variance rectification
    term; 'sgd': momentum-only update (only with degenerated_to_sgd); 'none': the parameters are left alone (moments still move)."""
    beta2_t = beta2 ** step
    n_sma_max = 2.0 / (1.0 - beta2) - 1.0
    n_sma = n_sma_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
    if n_sma >= 5:
        return "rect", math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (
            1 - beta1 ** step)
    if degenerated_to_sgd:
        return "sgd", 1.0 / (1 - beta1 ** step)
    return "none", -1.0


def radam_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float, weight_decay, degenerated_to_sgd: bool):
    """One RAdam step, in place on float64 arrays (nntrainer/optimization.py:112-181).  weight_decay may be an array
    (weight_decay * decay_mult per element, :66-72).  Decoupled decay: p -= wd * lr * p before the update."""
    v *= beta2; v += (1 - beta2) * g * g
    m *= beta1; m += (1 - beta1) * g
    mode, step_size = radam_scalars(step, beta1, beta2, degenerated_to_sgd)
    if mode == "rect":
        p -= weight_decay * lr * p
        p -= step_size * lr * m / (np.sqrt(v) + eps)
    elif mode == "sgd":
        p -= weight_decay * lr * p
        p -= step_size * lr * m
    return p


# ---------------------------------------------------------------------------------------------
# retrieval metrics
# ---------------------------------------------------------------------------------------------
def compute_retrieval_cosine(dot_product: np.ndarray) -> Tuple[Dict[str, float], np.ndarray]:
    """nntrainer/retrieval.py:68-98: rank of the diagonal item in argsort(row)[::-1]."""
    n = len(dot_product)
    ranks = np.empty(n)
    for i in range(n):
        inds = np.argsort(dot_product[i])[::-1]
        ranks[i] = np.where(inds == i)[0][0]
    r1 = float((ranks < 1).mean())
    r5 = float((ranks < 5).mean())
    r10 = float((ranks < 10).mean())
    r50 = float((ranks < 50).mean())
    medr = float(np.floor(np.median(ranks)) + 1)
    meanr = float(ranks.mean() + 1)
    return dict(r1=r1, r5=r5, r10=r10, r50=r50, medr=medr, meanr=meanr, sum=r1 + r5 + r50), ranks


def compute_retrieval(emb1: np.ndarray, emb2: np.ndarray):
    """nntrainer/retrieval.py:31-65 (embeddings already L2 normalised by the caller,
    coot/trainer_retrieval.py:397-402)."""
    d = emb1 @ emb2.T
    r12, _ = compute_retrieval_cosine(d)
    r21, _ = compute_retrieval_cosine(d.T)
    return r12, r21, (r12["r1"] + r21["r1"]) / 2


# ---------------------------------------------------------------------------------------------
# deterministic parameter / batch generators shared by fixtures, tests and the bench
# ---------------------------------------------------------------------------------------------
def param_shapes(cfg: NetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict names and shapes of one TransformerLegacy."""
    D, F = cfg.hidden_dim, cfg.ff_dim
    out: List[Tuple[str, Tuple[int, ...]]] = [("norm_input.gain", (cfg.input_dim,)),
                                               ("norm_input.bias", (cfg.input_dim,))]
    if cfg.use_input_fc:
        out += [("input_fc.mlp.0.weight", (D, cfg.input_dim)), ("input_fc.mlp.0.bias", (D,))]

    def layer(pre):
        a = pre + "self_attention_layer."
        f = pre + "pointwise_feedforward_layer."
        r = []
        for nm in ("query", "key", "value", "final"):
            r += [(a + f"sublayer.{nm}_projection.weight", (D, D)), (a + f"sublayer.{nm}_projection.bias", (D,))]
        r += [(a + "layer_normalization.gain", (D,)), (a + "layer_normalization.bias", (D,))]
        r += [(f + "sublayer.feed_forward.0.weight", (F, D)), (f + "sublayer.feed_forward.0.bias", (F,)),
              (f + "sublayer.feed_forward.3.weight", (D, F)), (f + "sublayer.feed_forward.3.bias", (D,)),
              (f + "layer_normalization.gain", (D,)), (f + "layer_normalization.bias", (D,))]
        return r

    for i in range(cfg.num_layers):
        out += layer(f"tf.encoder_layers.{i}.")
    if cfg.use_context:
        for i in range(cfg.ctx_num_layers):
            out += layer(f"tf_context.encoder_layers.{i}.")
    if cfg.pooler == "atn":
        H = cfg.pool_heads
        dh, do = cfg.pool_hidden // H, D // H
        out += [("pooler.pools.0.genpool_w1_head", (H, D, dh)), ("pooler.pools.0.genpool_b1_head", (H, dh)),
                ("pooler.pools.0.genpool_w2_head", (H, dh, do)), ("pooler.pools.0.genpool_b2_head", (H, do))]
    return out


def make_params(cfg: NetConfig, seed: int, scale: float = 0.05, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Seeded 'trained-like' parameters (np.random.RandomState: stream frozen across numpy
    versions).  LN gains ~ 1 +- 0.1, everything else N(0, scale) so that every path (biases, LN
    affine, pooling) is exercised; the reference's own init (truncnorm sigma=0.01) makes
    attention nearly uniform and hides bugs."""
    rs = np.random.RandomState(seed)
    P: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("gain"):
            v = 1.0 + 0.1 * rs.randn(*shape)
        elif "layer_normalization.bias" in name or name == "norm_input.bias":
            v = 0.1 * rs.randn(*shape)
        else:
            fan_in = shape[-1] if "genpool_w" not in name else shape[1]
            v = rs.randn(*shape) * (scale if len(shape) == 1 else 1.0 / math.sqrt(fan_in))
        P[name] = v.astype(dtype)
    P["embedding.pe"] = sincos_pe(1000, cfg.hidden_dim)
    return P


def make_batch(seed: int, B: int, counts, Lv: int, Lc: int, Lp: int, Ls: int, Dv: int, Dt: int,
               ragged: bool = True, dtype=np.float64, corr: float = 0.0):
    """Seeded synthetic RetrievalDataBatchTuple content (coot/dataset_retrieval.py:64-102 layout):
    zero padded features, lengths, counts.  counts: int (fixed clips per video) or sequence."""
    rs = np.random.RandomState(seed)
    counts = np.full(B, counts, dtype=np.int64) if np.isscalar(counts) else np.asarray(counts, dtype=np.int64)
    assert len(counts) == B
    Nc = int(counts.sum())

    def lens(n, L, lo):
        if not ragged:
            return np.full(n, L, dtype=np.int64)
        l = rs.randint(lo, L + 1, size=n).astype(np.int64)
        l[rs.randint(0, n)] = L  # at least one full-length row (collate pads to the max)
        return l

    def feats(n, L, D, ln):
        x = rs.randn(n, L, D)
        x[np.arange(L)[None, :] >= ln[:, None]] = 0.0
        return x.astype(dtype)

    vid_len, par_len = lens(B, Lv, max(1, Lv // 4)), lens(B, Lp, max(1, Lp // 4))
    clip_len, sent_len = lens(Nc, Lc, max(1, Lc // 8)), lens(Nc, Ls, max(1, Ls // 4))
    b = dict(vid_feat=feats(B, Lv, Dv, vid_len), vid_feat_len=vid_len,
             par_feat=feats(B, Lp, Dt, par_len), par_feat_len=par_len,
             clip_num=counts.copy(), clip_feat=feats(Nc, Lc, Dv, clip_len), clip_feat_len=clip_len,
             sent_num=counts.copy(), sent_feat=feats(Nc, Ls, Dt, sent_len), sent_feat_len=sent_len)
    if corr > 0:
        # correlated text features so retrieval is non-trivial: text = video-mean projected + noise
        Pm = rs.randn(Dv, Dt) / math.sqrt(Dv)
        for (vf, vl, tf, tl) in (("vid_feat", "vid_feat_len", "par_feat", "par_feat_len"),
                                 ("clip_feat", "clip_feat_len", "sent_feat", "sent_feat_len")):
            vm = b[vf].sum(1) / b[vl][:, None]
            t = b[tf]
            t += corr * (vm @ Pm)[:, None, :]
            t[np.arange(t.shape[1])[None, :] >= b[tl][:, None]] = 0.0
    for k in ("vid", "par", "clip", "sent"):
        L = b[f"{k}_feat"].shape[1]
        b[f"{k}_feat_mask"] = np.arange(L)[None, :] >= b[f"{k}_feat_len"][:, None]
    return b


def anet_like_counts(seed, B):
    """Clips per video with the shape of the ActivityNet annotation statistics (mean 3.74, p95 7, max 27)."""
    rs = np.random.RandomState(seed)
    return np.minimum(27, 1 + rs.negative_binomial(2, 0.42, B)).astype(np.int64)


def make_latent_batch(seed: int, B: int, counts, Lv: int, Lc: int, Lp: int, Ls: int, Dv: int, Dt: int, latent: int = 32,
                      noise: float = 0.5, map_seed: int = 12345, clusters: int = 0, spread: float = 1.0):
    """Seeded batch with a LEARNABLE video-text correspondence (SURVEY 8d retrieval-parity set): every clip / sentence pair shares
    a latent code u ~ N(0, I_latent); its frames are A u + noise, its words B u + noise (A, B fixed by map_seed, the same for
    every batch); the frames of a video / the words of its paragraph are those of its clips / sentences, sub-sampled to Lv / Lp.
    A few hundred contrastive steps make retrieval far better than chance, unlike make_batch's weak mean-feature coupling."""
    rs = np.random.RandomState(seed)
    rm = np.random.RandomState(map_seed)
    A = rm.randn(latent, Dv) / math.sqrt(latent)
    Bm = rm.randn(latent, Dt) / math.sqrt(latent)
    counts = np.full(B, counts, dtype=np.int64) if np.isscalar(counts) else np.asarray(counts, dtype=np.int64)
    Nc = int(counts.sum())
    u = rs.randn(Nc, latent)
    if clusters > 0:
        # confusable neighbours: every code sits near one of `clusters` shared centres (fixed by map_seed), `spread` of its norm
        # is its own — retrieval inside a cluster has to resolve the small individual part (near-ties, as in real data)
        centres = rm.randn(clusters, latent)
        u = math.sqrt(max(0.0, 1.0 - spread * spread)) * centres[rs.randint(0, clusters, size=Nc)] + spread * u
    clip_len = rs.randint(max(1, Lc // 4), Lc + 1, size=Nc).astype(np.int64); clip_len[rs.randint(0, Nc)] = Lc
    sent_len = rs.randint(max(1, Ls // 3), Ls + 1, size=Nc).astype(np.int64); sent_len[rs.randint(0, Nc)] = Ls

    def seqs(n, L, D, ln, base):
        x = base[:, None, :] + noise * rs.randn(n, L, D)
        x[np.arange(L)[None, :] >= ln[:, None]] = 0.0
        return x

    clip_feat = seqs(Nc, Lc, Dv, clip_len, u @ A)
    sent_feat = seqs(Nc, Ls, Dt, sent_len, u @ Bm)
    vid_len = np.minimum(Lv, np.maximum(1, [int(clip_len[s:s + c].sum()) for s, c in zip(np.cumsum(counts) - counts, counts)])).astype(np.int64)
    par_len = np.minimum(Lp, np.maximum(1, [int(sent_len[s:s + c].sum()) for s, c in zip(np.cumsum(counts) - counts, counts)])).astype(np.int64)
    vid_feat, par_feat = np.zeros((B, Lv, Dv)), np.zeros((B, Lp, Dt))
    ptr = 0
    for b, c in enumerate(counts):
        fr = np.concatenate([clip_feat[ptr + i, :clip_len[ptr + i]] for i in range(c)])
        wd = np.concatenate([sent_feat[ptr + i, :sent_len[ptr + i]] for i in range(c)])
        iv = np.linspace(0, len(fr) - 1, vid_len[b]).round().astype(int)
        ip = np.linspace(0, len(wd) - 1, par_len[b]).round().astype(int)
        vid_feat[b, :vid_len[b]] = fr[iv]
        par_feat[b, :par_len[b]] = wd[ip]
        ptr += c
    b_ = dict(vid_feat=vid_feat, vid_feat_len=vid_len, par_feat=par_feat, par_feat_len=par_len, clip_num=counts.copy(),
              clip_feat=clip_feat, clip_feat_len=clip_len, sent_num=counts.copy(), sent_feat=sent_feat, sent_feat_len=sent_len)
    for k in ("vid", "par", "clip", "sent"):
        L = b_[f"{k}_feat"].shape[1]
        b_[f"{k}_feat_mask"] = np.arange(L)[None, :] >= b_[f"{k}_feat_len"][:, None]
    return b_


# ---- input side: seeded data points for the collation tests (shared by oracle/gen_golden.py and tests/) ----------------
def make_datapoints(seed, B, dv, dt, max_frames=9, max_words=7, max_clips=4):
    """B synthetic videos in the layout of RetrievalDataset.__getitem__ (coot/dataset_retrieval.py:261-333): a dict per
    video with key, vid_feat [Lv, dv], clip_feat_list ([Lc_i, dv] per clip), par_feat [sum of sentence lengths, dt] and
    sent_feat_len_list — sentences are slices of the paragraph features there, so only the lengths are stored."""
    rs = np.random.RandomState(seed)
    pts = []
    for b in range(B):
        c = int(rs.randint(1, max_clips + 1))
        lv = int(rs.randint(1, max_frames + 1))
        clip_lens = [int(rs.randint(1, max_frames + 1)) for _ in range(c)]
        sent_lens = [int(rs.randint(1, max_words + 1)) for _ in range(c)]
        pts.append(dict(key=f"v_{seed}_{b}", vid_feat=rs.randn(lv, dv).astype(np.float32),
                        clip_feat_list=[rs.randn(n, dv).astype(np.float32) for n in clip_lens],
                        par_feat=rs.randn(sum(sent_lens), dt).astype(np.float32), sent_feat_len_list=sent_lens))
    return pts
