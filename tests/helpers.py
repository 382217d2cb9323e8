"""Shared helpers for the parity tests (HIP path vs oracle/golden)."""
import copy
import os

import numpy as np

from oracle import coot_oracle as O


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float -> bfloat16 bit pattern (uint16), RNE."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def cosine_rows(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = a.reshape(-1, a.shape[-1]).astype(np.float64)
    b = b.reshape(-1, b.shape[-1]).astype(np.float64)
    return (a * b).sum(-1) / np.maximum(np.sqrt((a * a).sum(-1) * (b * b).sum(-1)), 1e-30)


def cosine_flat(a: np.ndarray, b: np.ndarray) -> float:
    a, b = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
    return float((a * b).sum() / max(np.sqrt((a * a).sum() * (b * b).sum()), 1e-30))


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max abs error relative to the reference's scale (max |b|)."""
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def ocfg_to_dict(cfg: O.NetConfig, name="transformer_hip", dropout=0.0) -> dict:
    """oracle NetConfig -> one net_* section in the reference YAML schema."""
    d = dict(name=name, output_dim=cfg.hidden_dim * (2 if cfg.use_context else 1), use_input_fc=cfg.use_input_fc,
             positional_encoding="sincos", add_local_cls_token=False, dropout_input=0, norm_input="layernorm_coot",
             selfatn_config=dict(hidden_dim=cfg.hidden_dim, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                                 pointwise_ff_dim=cfg.ff_dim, activation="gelu", dropout=dropout, norm="layernorm_coot"),
             use_context=cfg.use_context, use_output_fc=False, weight_init_type="truncnorm", weight_init_std=0.01)
    if cfg.use_input_fc:
        d["input_fc_config"] = dict(output_dim=cfg.hidden_dim, num_layers=1, hidden_dim=0, activation_middle="none",
                                    activation_output="gelu", dropout_middle=0, dropout_output=0, norm_middle="none",
                                    norm_output="none", residual="none")
    if cfg.use_context:
        d["crossatn_config"] = dict(hidden_dim=cfg.hidden_dim, num_layers=cfg.ctx_num_layers, num_heads=cfg.num_heads,
                                    pointwise_ff_dim=cfg.ff_dim, activation="gelu", dropout=dropout, norm="layernorm_coot")
    if cfg.pooler == "atn":
        d["pooler_config"] = dict(name="atn", hidden_dim=cfg.pool_hidden, num_heads=cfg.pool_heads, num_layers=1,
                                  dropout=dropout, activation="gelu")
    else:
        d["pooler_config"] = dict(name="avg_special")
    return d


def make_hip_net(cfg: O.NetConfig, P: dict, dropout=0.0, device="cuda"):
    import torch
    import coot_videotext_amd as cva
    tc = cva.TransformerConfig(ocfg_to_dict(cfg, dropout=dropout), cfg.input_dim)
    net = cva.TransformerHip(tc)
    sd = net.state_dict()
    for k in sd:
        if k in P:
            sd[k] = torch.from_numpy(np.asarray(P[k], dtype=np.float32))
    net.load_state_dict(sd)
    return net.to(device)


def full_cfgs(dv, dt, hidden, heads, ff, ph, layers=1):
    """layers: encoder layers of the LOCAL networks (the global ones keep one: BASELINE.json configs[0])."""
    kw = dict(hidden_dim=hidden, num_heads=heads, ff_dim=ff, pool_hidden=ph)
    return [O.NetConfig(input_dim=dv, num_layers=layers, **kw),
            O.NetConfig(input_dim=hidden, use_input_fc=False, use_context=True, pooler="avg_special", **kw),
            O.NetConfig(input_dim=dt, num_layers=layers, **kw),
            O.NetConfig(input_dim=hidden, use_input_fc=False, use_context=True, pooler="avg_special", **kw)]


NET_KEYS = ["net_video_local", "net_video_global", "net_text_local", "net_text_global"]
ANET_W = dict(weight_high=1.0, weight_high_internal=1.0, weight_low=1.0, weight_low_internal=1.0,
              weight_context=1.0, weight_context_internal=0.0)


def make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.01, device="cuda", optimizer=None):
    """RetrievalModelManager + RetrievalTrainer for four oracle configs/param sets.  optimizer: overrides of the optimizer section."""
    import torch
    import coot_videotext_amd as cva
    raw = dict(train=dict(batch_size=4, loss_func="contrastive", contrastive_loss_config=dict(margin=0.2, **ANET_W),
                          loss_cycle_cons=cc_weight),
               dataset_train=dict(vid_feat_dim=cfgs[0].input_dim, text_feat_dim=cfgs[2].input_dim),
               optimizer=dict(name="adam", lr=1e-3, weight_decay=2e-5, weight_decay_for_bias=True, momentum=0.9,
                              adam_beta2=0.999, adam_eps=1e-8),
               use_cuda=True, fp16_train=True, fp16_val=True)
    raw["optimizer"].update(optimizer or {})
    for k, c in zip(NET_KEYS, cfgs):
        raw[k] = ocfg_to_dict(c, dropout=dropout)
    cfg = cva.RetrievalConfig(raw)
    mgr = cva.RetrievalModelManager(cfg)
    for k, P in zip(NET_KEYS, Ps):
        sd = mgr.model_dict[k].state_dict()
        for n in sd:
            if n in P:
                sd[n] = torch.from_numpy(np.asarray(P[n], dtype=np.float32))
        mgr.model_dict[k].load_state_dict(sd)
    mgr.cuda()
    return cfg, mgr


def oracle_full(cfgs, Ps, b, idx_clip, idx_sent, q=O.EXACT, w=ANET_W, margin=0.2, cc_weight=0.01, bwd=True):
    vis, cv = O.encode_side(Ps[0], cfgs[0], Ps[1], cfgs[1], b["vid_feat"], b["vid_feat_len"], b["clip_feat"],
                            b["clip_feat_len"], b["clip_num"], q)
    txt, ct = O.encode_side(Ps[2], cfgs[2], Ps[3], cfgs[3], b["par_feat"], b["par_feat_len"], b["sent_feat"],
                            b["sent_feat_len"], b["sent_num"], q)
    E = dict(vid_emb=vis["global_emb"], par_emb=txt["global_emb"], clip_emb=vis["item_emb"],
             sent_emb=txt["item_emb"], vid_context=vis["context"], par_context=txt["context"])
    contr, dE = O.total_contrastive_loss(E, w, margin, q)
    cvalid, svalid = ~vis["item_emb_mask"], ~txt["item_emb_mask"]
    lc, ls = O.cycle_consistency_loss(vis["item_emb_reshape"], cvalid, txt["item_emb_reshape"], svalid, idx_clip, idx_sent)
    cc = cc_weight * (lc + ls)
    Gs = None
    if bwd:
        dcr, dsr = O.cycle_consistency_bwd(vis["item_emb_reshape"], cvalid, txt["item_emb_reshape"], svalid,
                                           idx_clip, idx_sent, cc_weight)
        Gvl, Gvg = O.encode_side_bwd(Ps[0], cfgs[0], Ps[1], cfgs[1], cv, dE["vid_emb"], dE["clip_emb"], dE["vid_context"], dcr)
        Gtl, Gtg = O.encode_side_bwd(Ps[2], cfgs[2], Ps[3], cfgs[3], ct, dE["par_emb"], dE["sent_emb"], dE["par_context"], dsr)
        Gs = [Gvl, Gvg, Gtl, Gtg]
    return vis, txt, contr, cc, Gs


def grad_report(named_got, ref: dict, cos_min=0.99, ratio_tol=0.05):
    """Compare parameter gradients with a reference.  Parameters whose true gradient is (numerically) zero
    (e.g. key-projection bias, 2nd pooling bias: softmax shift invariance) are checked by magnitude only.
    Returns (bad list, table string)."""
    rows, bad = [], []
    scale = max(float(np.linalg.norm(np.asarray(v, np.float64))) for v in ref.values())
    for name, g in named_got:
        r = np.asarray(ref[name], np.float64)
        g = np.asarray(g, np.float64)
        nr, ng = float(np.linalg.norm(r)), float(np.linalg.norm(g))
        if nr < 1e-6 * scale:
            ok = ng < 1e-3 * scale
            rows.append(f"{name:90s} zero-grad  |got|={ng:.2e} |ref|={nr:.2e} {'ok' if ok else 'BAD'}")
        else:
            c = cosine_flat(g, r)
            ok = c > cos_min and abs(ng / nr - 1) < ratio_tol
            rows.append(f"{name:90s} cos={c:.5f} ratio={ng / nr:.4f} {'ok' if ok else 'BAD'}")
        if not ok:
            bad.append(rows[-1])
    return bad, "\n".join(rows)


def rank_flips(e1, e2, r1, r2):
    """Retrieval ranks (nntrainer/retrieval.py:68-98: number of items scored above the paired one) of the L2-normalised embedding
    sets (e1, e2) against those of the reference's (r1, r2), both directions.  Returns (number of (query, item) comparisons whose
    outcome differs, the largest REFERENCE margin |S_ij - S_ii| among them): a flip with a margin far above the embedding
    tolerance would be a real ranking difference, flips inside it are the near-ties any two fp32 implementations disagree on."""
    def nrm(x):
        x = np.asarray(x, np.float64)
        return x / np.sqrt((x * x).sum(-1, keepdims=True))
    S, R = nrm(e1) @ nrm(e2).T, nrm(r1) @ nrm(r2).T
    flips, worst = 0, 0.0
    for A, B in ((S, R), (S.T, R.T)):
        da, db = A - np.diag(A)[:, None], B - np.diag(B)[:, None]
        diff = (da > 0) != (db > 0)
        np.fill_diagonal(diff, False)
        flips += int(diff.sum())
        if diff.any():
            worst = max(worst, float(np.abs(db[diff]).max()))
    return flips, worst


def import_reference(ref_root=None):
    """The unmodified reference, imported with the shims of SURVEY 8c (removed collections aliases, absent GPUtil / h5py /
    tensorboard).  Returns a namespace of its modules, or None where the reference tree does not exist (the GPU box)."""
    import collections
    import collections.abc
    import sys
    import types
    if ref_root is None:  # COOT_REFERENCE_ROOT: a copy of the reference's coot/ + nntrainer/ + config/ shipped to the GPU box for one run
        ref_root = os.environ.get("COOT_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "coot")):
        return None
    for n in ("Iterable", "Mapping", "Sequence", "MutableMapping"):
        setattr(collections, n, getattr(collections.abc, n))
    for name in ("GPUtil", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"add_scalar": lambda *a, **k: None, "close": lambda *a, **k: None})
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    import warnings
    warnings.filterwarnings("ignore")
    from coot import configs_retrieval, model_retrieval
    from nntrainer import models, utils_yaml
    return types.SimpleNamespace(root=ref_root, model_retrieval=model_retrieval, configs_retrieval=configs_retrieval, models=models,
                                 utils_yaml=utils_yaml)
