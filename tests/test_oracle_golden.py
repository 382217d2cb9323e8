"""Pins oracle/coot_oracle.py (numpy restatement) against fixtures produced by the unmodified
reference (oracle/gen_golden.py -> tests/golden/*.npz).  CPU only."""
import copy
import os

import numpy as np
import pytest

from oracle import coot_oracle as O

NET_KEYS = ["net_video_local", "net_video_global", "net_text_local", "net_text_global"]


def _close(a, b, rtol=2e-4, atol=2e-5, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (ref scale {np.abs(b).max():.3e})"


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


SINGLE = {
    "net_local_small": (O.NetConfig(input_dim=40, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64), 11, False),
    "net_global_small": (O.NetConfig(input_dim=32, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64,
                                     use_input_fc=False, use_context=True, pooler="avg_special"), 13, True),
    "net_local_2layer": (O.NetConfig(input_dim=40, hidden_dim=32, num_heads=4, ff_dim=48, pool_hidden=32,
                                     num_layers=2), 17, False),
}


@pytest.mark.parametrize("name", list(SINGLE))
def test_single_net_fwd_bwd(golden_dir, name):
    cfg, seed, with_ctx = SINGLE[name]
    g = _load(golden_dir, name)
    P = O.make_params(cfg, seed)
    x = g["x"].astype(np.float64)
    hid = g["hidden"].astype(np.float64) if with_ctx else None
    pooled, per_tok, cache = O.net_fwd(P, cfg, x, g["lens"], hid)
    _close(pooled, g["pooled"], what="pooled")
    _close(per_tok, g["per_token"], what="per_token")
    G, dhid, dx = O.net_bwd(P, cfg, g["R"].astype(np.float64), cache, need_dfeats=True)
    if with_ctx:
        _close(dhid, g["dhidden"], what="dhidden")
    # dx of rows whose input row is all-zero padding is a dead end (sigma = 0)
    valid = np.arange(x.shape[1])[None, :] < g["lens"][:, None]
    _close(dx[valid], g["dx"][valid], rtol=1e-3, atol=1e-4, what="dx")
    n = 0
    for k, v in g.items():
        if k.startswith("grad:"):
            _close(G[k[5:]], v, rtol=1e-3, atol=1e-4, what=k)
            n += 1
    assert n == len(G), (n, sorted(G))


def _full_setup(g):
    seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
    kw = dict(hidden_dim=hidden, num_heads=heads, ff_dim=ff, pool_hidden=ph)
    cfgs = [O.NetConfig(input_dim=dv, **kw),
            O.NetConfig(input_dim=hidden, use_input_fc=False, use_context=True, pooler="avg_special", **kw),
            O.NetConfig(input_dim=dt, **kw),
            O.NetConfig(input_dim=hidden, use_input_fc=False, use_context=True, pooler="avg_special", **kw)]
    Ps = [O.make_params(cfgs[i], seed + 10 * i) for i in range(4)]
    b = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=True, corr=0.5)
    return cfgs, Ps, b


ANET_W = dict(weight_high=1.0, weight_high_internal=1.0, weight_low=1.0, weight_low_internal=1.0,
              weight_context=1.0, weight_context_internal=0.0)


def oracle_full(cfgs, Ps, b, idx_clip, idx_sent, q=O.EXACT, w=ANET_W, margin=0.2, cc_weight=0.01):
    vis, cv = O.encode_side(Ps[0], cfgs[0], Ps[1], cfgs[1], b["vid_feat"], b["vid_feat_len"], b["clip_feat"],
                            b["clip_feat_len"], b["clip_num"], q)
    txt, ct = O.encode_side(Ps[2], cfgs[2], Ps[3], cfgs[3], b["par_feat"], b["par_feat_len"], b["sent_feat"],
                            b["sent_feat_len"], b["sent_num"], q)
    E = dict(vid_emb=vis["global_emb"], par_emb=txt["global_emb"], clip_emb=vis["item_emb"],
             sent_emb=txt["item_emb"], vid_context=vis["context"], par_context=txt["context"])
    contr, dE = O.total_contrastive_loss(E, w, margin, q)
    cvalid, svalid = ~vis["item_emb_mask"], ~txt["item_emb_mask"]
    lc, ls = O.cycle_consistency_loss(vis["item_emb_reshape"], cvalid, txt["item_emb_reshape"], svalid,
                                      idx_clip, idx_sent)
    cc = cc_weight * (lc + ls)
    dcr, dsr = O.cycle_consistency_bwd(vis["item_emb_reshape"], cvalid, txt["item_emb_reshape"], svalid,
                                       idx_clip, idx_sent, cc_weight)
    Gvl, Gvg = O.encode_side_bwd(Ps[0], cfgs[0], Ps[1], cfgs[1], cv, dE["vid_emb"], dE["clip_emb"],
                                 dE["vid_context"], dcr)
    Gtl, Gtg = O.encode_side_bwd(Ps[2], cfgs[2], Ps[3], cfgs[3], ct, dE["par_emb"], dE["sent_emb"],
                                 dE["par_context"], dsr)
    return vis, txt, contr, cc, [Gvl, Gvg, Gtl, Gtg]


def _check_embs(vis, txt, g, rtol=3e-4, atol=3e-5):
    _close(vis["global_emb"], g["vid_emb"], rtol, atol, "vid_emb")
    _close(vis["item_emb"], g["clip_emb"], rtol, atol, "clip_emb")
    _close(vis["context"], g["vid_context"], rtol, atol, "vid_context")
    _close(vis["item_emb_reshape"], g["clip_emb_reshape"], rtol, atol, "clip_emb_reshape")
    assert (vis["item_emb_mask"] == g["clip_emb_mask"]).all()
    assert (vis["item_emb_lens"] == g["clip_emb_lens"]).all()
    _close(txt["global_emb"], g["par_emb"], rtol, atol, "par_emb")
    _close(txt["item_emb"], g["sent_emb"], rtol, atol, "sent_emb")
    _close(txt["context"], g["par_context"], rtol, atol, "par_context")
    assert (txt["item_emb_mask"] == g["sent_emb_mask"]).all()


def test_full_small(golden_dir):
    g = _load(golden_dir, "full_small")
    cfgs, Ps, b = _full_setup(g)
    vis, txt, contr, cc, Gs = oracle_full(cfgs, Ps, b, g["cc_idx_clip"], g["cc_idx_sent"])
    _check_embs(vis, txt, g)
    _close(contr, g["contr_loss"], 1e-4, 1e-6, "contr")
    _close(cc, g["cc_loss"], 1e-3, 1e-7, "cc")
    rows = O.cycle_consistency_rows(vis["item_emb_reshape"], ~vis["item_emb_mask"], txt["item_emb_reshape"],
                                    ~txt["item_emb_mask"])
    _close(rows, np.where(~g["clip_emb_mask"], g["cc_rows_clip"], 0.0), 1e-3, 1e-5, "cc_rows_clip")
    rows = O.cycle_consistency_rows(txt["item_emb_reshape"], ~txt["item_emb_mask"], vis["item_emb_reshape"],
                                    ~vis["item_emb_mask"])
    _close(rows, np.where(~g["sent_emb_mask"], g["cc_rows_sent"], 0.0), 1e-3, 1e-5, "cc_rows_sent")
    n = 0
    for i, k in enumerate(NET_KEYS):
        for name, v in Gs[i].items():
            _close(v, g[f"grad:{k}:{name}"], rtol=2e-3, atol=2e-5, what=f"{k}:{name}")
            n += 1
    assert n == sum(1 for k in g if k.startswith("grad:"))


def test_full_anet_dims(golden_dir):
    g = _load(golden_dir, "full_anet")
    cfgs, Ps, b = _full_setup(g)
    vis, txt, contr, cc, Gs = oracle_full(cfgs, Ps, b, g["cc_idx_clip"], g["cc_idx_sent"])
    _check_embs(vis, txt, g, rtol=5e-4, atol=5e-5)
    _close(contr, g["contr_loss"], 1e-4, 1e-6, "contr")
    _close(cc, g["cc_loss"], 2e-3, 1e-7, "cc")
    for i, k in enumerate(NET_KEYS):
        for name, v in Gs[i].items():
            gn = float(g[f"gnorm:{k}:{name}"])
            assert abs(np.linalg.norm(v) - gn) <= 2e-3 * gn + 1e-6, (k, name, np.linalg.norm(v), gn)
            ref = g[f"gsub:{k}:{name}"]  # vectors are stored in full, matrices as a strided sample (gen_golden.py: gen_full)
            sub = v.reshape(-1)[::(1 if ref.size == v.size else int(g["sub_step"]))]
            _close(sub, ref, rtol=5e-3, atol=2e-3 * gn / np.sqrt(v.size) + 1e-7,
                   what=f"{k}:{name}")
    # retrieval metrics of these embeddings
    for (a, c, tag) in ((vis["global_emb"], txt["global_emb"], "vp"), (vis["item_emb"], txt["item_emb"], "cs")):
        e1 = a / np.sqrt((a ** 2).sum(-1, keepdims=True))
        e2 = c / np.sqrt((c ** 2).sum(-1, keepdims=True))
        r12, r21, s1 = O.compute_retrieval(e1, e2)
        got = [r12[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")] + \
              [r21[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")] + [s1]
        _close(got, g[f"ret_{tag}"], 1e-6, 1e-9, f"ret_{tag}")


def test_bf16_emulation_stays_within_north_star_tolerance(golden_dir):
    """bf16 rounding at the HIP dataflow's rounding points keeps embeddings within 1e-3 cosine of
    the reference (north_star tolerance)."""
    g = _load(golden_dir, "full_anet")
    cfgs, Ps, b = _full_setup(g)
    vis, txt, contr, cc, _ = oracle_full(cfgs, Ps, b, g["cc_idx_clip"], g["cc_idx_sent"], q=O.BF16)
    for got, key in ((vis["global_emb"], "vid_emb"), (vis["item_emb"], "clip_emb"), (vis["context"], "vid_context"),
                     (txt["global_emb"], "par_emb"), (txt["item_emb"], "sent_emb"), (txt["context"], "par_context")):
        ref = g[key].astype(np.float64)
        cos = (got * ref).sum(-1) / np.sqrt((got ** 2).sum(-1) * (ref ** 2).sum(-1))
        assert cos.min() > 1 - 1e-3, (key, cos.min())
    assert abs(contr - float(g["contr_loss"])) < 2e-2


def test_retrieval_metrics(golden_dir):
    g = _load(golden_dir, "retrieval_metrics")
    for i in range(3):
        res, ranks = O.compute_retrieval_cosine(g[f"d{i}"])
        assert (ranks == g[f"ranks{i}"]).all()
        _close([res[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")], g[f"res{i}"], 1e-9, 1e-12)


def test_radam_oracle_matches_reference(golden_dir):
    """oracle radam_step against the trajectory of the reference's in-file RAdam (nntrainer/optimization.py:79-181),
    degenerated_to_sgd off (the paper configs) and on; both phases (N_sma < 5: no update / SGD, N_sma >= 5: rectified)."""
    g = _load(golden_dir, "radam")
    lr, b1, b2, eps, wd = (float(g[k]) for k in ("lr", "beta1", "beta2", "eps", "wd"))
    n = len(g["p0"])
    wdv = np.where(np.arange(n) < 200, wd, 0.0)
    modes = set()
    for degen in (0, 1):
        p = g["p0"].astype(np.float64); m = np.zeros(n); v = np.zeros(n)
        for s in range(len(g["grads"])):
            O.radam_step(p, g["grads"][s].astype(np.float64), m, v, s + 1, lr, b1, b2, eps, wdv, bool(degen))
            modes.add(O.radam_scalars(s + 1, b1, b2, bool(degen))[0])
            _close(p, g[f"traj_degen{degen}"][s], 2e-6, 2e-8, what=f"radam degen={degen} step {s + 1}")
    assert modes == {"rect", "sgd", "none"}


def test_mask_semantics(golden_dir):
    """tests_nntrainer/test_transformers.py:22-79 in numeric form."""
    g = _load(golden_dir, "mask_semantics")
    cfg = O.NetConfig(input_dim=24, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64)
    P = O.make_params(cfg, 3)
    valid = np.arange(6)[None, :] < g["lens"][:, None]
    y0, _ = O.encoder_layer_fwd(P, "tf.encoder_layers.0.", g["x"].astype(np.float64), g["x"].astype(np.float64),
                                valid, 4)
    x2 = g["x2"].astype(np.float64)
    y1, _ = O.encoder_layer_fwd(P, "tf.encoder_layers.0.", x2, x2, valid, 4)
    _close(y0, g["y0"], what="y0")
    _close(y1, g["y1"], what="y1")
    assert np.abs(y0[valid] - y1[valid]).max() < 1e-9  # un-masked rows unaffected by masked inputs


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -8, -2.5, 3.1415927, 1e-30, 65504.0],
                 dtype=np.float32)
    import torch
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert (O.bf16_round(x) == ref).all()


def test_torch_cpu_port_matches_numpy_oracle():
    """oracle/coot_torch_cpu.py (the PyTorch-CPU restatement timed as bench.py's cpu_baseline) against the numpy oracle,
    which the tests above pin to the reference-generated fixtures: embeddings, both losses and every parameter
    gradient of the full path (ragged batch, cycle loss on)."""
    from oracle import coot_torch_cpu as T
    from tests import helpers as H
    dims = (40, 32, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 21 + i) for i in range(4)]
    counts = [2, 1, 3, 2]
    b = O.make_batch(5, 4, counts, 11, 9, 8, 6, dims[0], dims[1], ragged=True)
    ic, isent = np.array([1, 0, 2, 0]), np.array([0, 0, 1, 1])
    vis_o, txt_o, contr_o, cc_o, Gs = H.oracle_full(cfgs, Ps, b, ic, isent)
    Pt = [T.to_torch_params(P) for P in Ps]
    vis, txt, contr, cc = T.full_step(cfgs, Pt, b, ic, isent, H.ANET_W, 0.2, 0.01)
    for k in ("global_emb", "item_emb", "context"):
        assert np.abs(vis[k].detach().numpy() - vis_o[k]).max() < 2e-4 * max(1.0, np.abs(vis_o[k]).max())
        assert np.abs(txt[k].detach().numpy() - txt_o[k]).max() < 2e-4 * max(1.0, np.abs(txt_o[k]).max())
    assert abs(float(contr) - contr_o) < 1e-4 * max(1.0, abs(contr_o))
    assert abs(float(cc) - cc_o) < 1e-4 * max(1e-3, abs(cc_o))
    for P, G in zip(Pt, Gs):
        bad, table = H.grad_report([(n, P[n].grad.numpy()) for n in G if P[n].grad is not None], G, cos_min=0.9999, ratio_tol=1e-3)
        assert not bad, "\n".join(bad)


def test_committed_fixtures_regenerate_from_the_reference(golden_dir, tmp_path):
    """The pin of the pin: where the unmodified reference is present (the build container: /root/reference or COOT_REFERENCE), the
    committed generator re-run into a scratch directory reproduces the committed fixtures BIT FOR BIT — the end-to-end eval
    fixture, a train-mode fixture with the library's dropout masks injected into the reference's nn.Dropout sites, the 8-step training
    trajectories of the reference's step body + optimizer (round 5), and the host-side
    ones (collate, LR schedules, RAdam, masks, retrieval metrics).  On the GPU box the reference does not exist: skipped."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.environ.get("COOT_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "coot")):
        pytest.skip("no reference checkout here")
    names = ["full_small", "bench_yc2_100m_2layer_train", "traj_small", "traj_small_eps", "collate", "lr_schedule", "radam", "mask_semantics", "retrieval_metrics"]
    env = dict(os.environ, COOT_GOLDEN_OUT=str(tmp_path), COOT_REFERENCE=ref)
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "gen_golden.py")] + names, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in names:
        new, old = np.load(os.path.join(str(tmp_path), n + ".npz")), np.load(os.path.join(golden_dir, n + ".npz"))
        assert sorted(new.files) == sorted(old.files), n
        for k in new.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (n, k)
            assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b), (n, k)
