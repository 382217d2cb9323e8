"""The fp32 REFERENCE MODE of the library (coot_net_config.dtype = COOT_DTYPE_F32) against fixtures the unmodified
reference wrote: with every activation, weight and accumulation in fp32 the embeddings must agree with the reference's to fp32
round-off — where the bf16 fast path agrees to ~1e-3 of the output scale.  Turns "is this difference bf16 noise or a logic error?"
into a measurement: a wrong mask, position, residual or pooling rule shows up at 1e-2 ... 1 in BOTH modes, bf16 rounding only in one.
Round 5: the mode has a backward pass too (the derivative of the same op sequence in fp32): all parameter gradients of the eval-mode
fixtures to <= 1e-4 relative, with the bf16 path's gradient error reported as a multiple of it.
"""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

# eval-mode fixtures (oracle/gen_golden.py): a 6-video batch at the paper dims, the benchmark shapes (fixed, ragged, Cmax = 64, 2-layer local networks)
CASES = ["full_anet", "bench_anet", "bench_anet_ragged", "bench_hbm_stress", "bench_yc2_100m_2layer"]
KEYS = ("vid_emb", "clip_emb", "vid_context", "par_emb", "sent_emb", "par_context")


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _load(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
    cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph, layers=int(g["layers"]) if "layers" in g else 1)
    Ps = [O.make_params(cfgs[i], seed + 10 * i, scale=float(g["param_scale"])) for i in range(4)]
    b = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=bool(int(g["ragged"])), corr=0.5)
    return g, cfgs, Ps, b


def _embeddings(torch, mgr, batch):
    with torch.no_grad():
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
    torch.cuda.synchronize()
    return {"vid_emb": vis.vid_emb, "clip_emb": vis.clip_emb, "vid_context": vis.vid_context, "par_emb": txt.par_emb, "sent_emb": txt.sent_emb,
            "par_context": txt.par_context}


@pytest.mark.parametrize("name", CASES)
def test_f32_reference_mode_matches_the_reference_to_round_off(env, golden_dir, name):
    torch, cva = env
    g, cfgs, Ps, b = _load(golden_dir, name)
    cfg, mgr = H.make_manager(cfgs, Ps, cc_weight=float(g["cc_weight"]) if "cc_weight" in g else 0.01)
    mgr.set_all_models_eval()
    batch = cva.synthetic.batch_from_numpy(b)
    fast = {k: v.cpu().numpy().astype(np.float64) for k, v in _embeddings(torch, mgr, batch).items()}
    for net in mgr.model_dict.values():
        net.set_compute_dtype("f32")
    ref = {k: v.cpu().numpy().astype(np.float64) for k, v in _embeddings(torch, mgr, batch).items()}
    worst_abs, worst_fast = 0.0, 0.0
    for k in KEYS:
        want = g[k].astype(np.float64)
        scale = max(1.0, float(np.abs(want).max()))
        e32, e16 = float(np.abs(ref[k] - want).max()), float(np.abs(fast[k] - want).max())
        worst_abs, worst_fast = max(worst_abs, e32 / scale), max(worst_fast, e16 / scale)
        print(f"[{name}] {k}: max |f32 mode - reference| = {e32:.2e}, max |bf16 path - reference| = {e16:.2e} (|reference| max {np.abs(want).max():.3f})")
        assert e32 <= 1e-5 * scale, (k, e32)
    # the two modes differ by the bf16 rounding of the fast path and by nothing else
    assert worst_fast > 10 * worst_abs, (worst_fast, worst_abs)
    # train mode must refuse (the checker has no dropout), not silently run the bf16 kernels
    net = mgr.model_dict["net_video_global"]
    x = torch.randn(2, 3, cfgs[1].input_dim, device="cuda")
    net.train()
    with pytest.raises(RuntimeError, match="eval-mode"):
        net(x, None, torch.tensor([3, 2], device="cuda"), torch.randn(2, cfgs[1].hidden_dim, device="cuda"))


def _grads_vs_fixture(g, mgr, tag):
    """Per tensor: relative L2 error of the (sub-sampled) gradient against the reference's; returns (worst, number checked)."""
    step = int(g["sub_step"]) if "sub_step" in g else 97
    gmax = max(float(g[k]) for k in g if k.startswith("gnorm:"))
    worst, checked, rows = 0.0, 0, []
    for k in H.NET_KEYS:
        net = mgr.model_dict[k]
        for pname, prm in net.named_parameters():
            key = f"{k}:{pname}"
            if "gnorm:" + key not in g:
                continue
            got = prm.grad.detach().cpu().numpy().reshape(-1).astype(np.float64) if prm.grad is not None else np.zeros(prm.numel())
            gn = float(g["gnorm:" + key])
            if gn < 1e-6 * gmax:  # (a zero gradient up to rounding: the key bias under a softmax)
                assert np.linalg.norm(got) < 1e-4 * gmax, (tag, key, float(np.linalg.norm(got)))
                continue
            ref = g["gsub:" + key].astype(np.float64)
            sub = got[::(1 if ref.size == got.size else step)]
            err = float(np.linalg.norm(sub - ref) / max(np.linalg.norm(ref), 1e-30))
            nr = float(np.linalg.norm(got)) / gn
            rows.append((err, abs(nr - 1), key))
            worst = max(worst, err, abs(nr - 1))
            checked += 1
    rows.sort(reverse=True)
    for err, ne, key in rows[:4]:
        print(f"[{tag}]   {key}: relative error {err:.2e}, norm error {ne:.2e}")
    return worst, checked


# (ragged batches, the fixed benchmark shape, two encoder layers per local network, Cmax = 64)
@pytest.mark.parametrize("name", ["full_anet", "bench_anet", "bench_anet_ragged", "bench_yc2_100m_2layer", "bench_hbm_stress"])
def test_f32_reference_mode_gradients_match_the_reference(env, golden_dir, name):
    """The library END TO END in its fp32 reference mode against the parameter gradients the unmodified reference wrote (eval-mode
    fixtures, dropout off): every one of the 108+ non-zero gradients to <= 1e-4 relative — the bf16 path's bound is cosine > 0.999 /
    norm within 1 %, i.e. its error is reported below as a multiple of this mode's.
      encoders: TransformerHip in f32 mode through torch.autograd (coot_net_fwd / coot_net_bwd, csrc/ref_f32.hip)
      losses:   the trainer's own hooks — coot_contrastive_fwd_bwd_f32 (csrc/loss_f32.hip; round 6: rounds 4-5 borrowed the torch
                restatement of oracle/coot_torch_cpu.py here because the fused loss computes similarities on the bf16 MFMA) and
                coot_cyclecons_fwd_bwd (fp32 VALU code in both modes), with the fixture's th.multinomial draws injected.
    Nothing under oracle/ takes part in the forward or backward of this test (it only rebuilds the fixture's inputs)."""
    torch, cva = env
    g, cfgs, Ps, b = _load(golden_dir, name)
    cc_w = float(g["cc_weight"]) if "cc_weight" in g else 0.01
    cfg, mgr = H.make_manager(cfgs, Ps, cc_weight=cc_w)
    mgr.set_all_models_eval()
    trainer = cva.RetrievalTrainer(cfg, mgr, is_test=True)
    batch = cva.synthetic.batch_from_numpy(b)
    idx_c = torch.from_numpy(g["cc_idx_clip"].astype(np.int64)).cuda()
    idx_s = torch.from_numpy(g["cc_idx_sent"].astype(np.int64)).cuda()

    def run(mode):
        for net in mgr.model_dict.values():
            net.set_compute_dtype(mode)
            for prm in net.parameters():
                prm.grad = None
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
        contr = trainer.compute_total_constrastive_loss(vis, txt)  # (follows the networks' mode: f32 -> coot_contrastive_fwd_bwd_f32)
        cc = trainer.compute_cyclecons_loss(vis, txt, idx_c, idx_s)
        (contr + cc).backward()
        torch.cuda.synchronize()
        return float(contr.detach()), float(cc.detach())

    contr, cc = run("f32")
    rc, rcc = float(g["contr_loss"]), float(g["cc_loss"])
    print(f"[{name}] f32 mode: contrastive {contr:.7f} vs {rc:.7f}; cycle-consistency {cc:.8f} vs {rcc:.8f}")
    assert abs(contr - rc) < 1e-5 * abs(rc) and abs(cc - rcc) < 1e-4 * abs(rcc) + 1e-9
    worst32, n32 = _grads_vs_fixture(g, mgr, name + " f32")
    print(f"[{name}] f32 mode: {n32} parameter gradients, worst relative error {worst32:.2e}")
    assert n32 >= 100 and worst32 <= 1e-4, worst32
    run("bf16")
    worst16, _ = _grads_vs_fixture(g, mgr, name + " bf16")
    print(f"[{name}] bf16 path (fused bf16 losses): worst relative error {worst16:.2e} = {worst16 / max(worst32, 1e-12):.0f} x the f32 mode's")
    assert worst16 > 10 * worst32


def test_f32_loss_mode_matches_torch_ops_and_the_step_api_refuses_f32(env):
    """coot_contrastive_fwd_bwd_f32 on its own against the same loss written with ATen fp32 ops ON THE GPU (F.normalize, the seven
    ContrastiveLoss terms of coot/trainer_retrieval.py:148-182 incl. the :181 weight quirk, autograd): loss and all six gradients to
    fp32 round-off, for the shipped weights and for a set that switches every term on; and the ADVICE round-5 guard — the step API
    refuses networks in the f32 checker mode instead of accumulating their weight-matrix gradients across steps."""
    torch, cva = env
    import torch.nn.functional as F
    gen = torch.Generator(device="cuda").manual_seed(5)
    nh, nl, dh, dl = 24, 70, 96, 40

    def contr(a, b_, m):
        sc = a @ b_.t()
        d = sc.diag().view(-1, 1)
        eye = torch.eye(sc.shape[0], dtype=torch.bool, device=sc.device)
        return ((m + sc - d).clamp(min=0).masked_fill(eye, 0).sum() + (m + sc - d.t()).clamp(min=0).masked_fill(eye, 0).sum()) / sc.shape[0] ** 2

    for w in (H.ANET_W, dict(weight_high=0.7, weight_high_internal=0.3, weight_low=1.1, weight_low_internal=0.6, weight_context=0.9,
                             weight_context_internal=1.0)):
        def rows(n, d):  # a component common to all rows: negatives at cosine ~0.67, partners at ~0.92 -> some hinge terms violated, some not
            return torch.randn(1, d, device="cuda", generator=gen) + 0.7 * torch.randn(n, d, device="cuda", generator=gen)
        sets = [rows(nh, dh), None, rows(nl, dl), None, rows(nh, dl), None]
        for k in (1, 3, 5):
            sets[k] = sets[k - 1] + 0.5 * torch.randn(sets[k - 1].shape, device="cuda", generator=gen)
        ref_in = [t.clone().double().requires_grad_(True) for t in sets]
        ve, pe, ce, se, vc, pc = [F.normalize(t) for t in ref_in]
        loss_ref = 0
        for wt, a, b_ in ((w["weight_high"], ve, pe), (w["weight_low"], ce, se), (w["weight_context"], vc, pc)):
            loss_ref = loss_ref + wt * contr(a, b_, 0.2)
        for wt, a, b_ in ((w["weight_high_internal"], ve, pe), (w["weight_low_internal"], ce, se),
                          (w["weight_low_internal"] if w["weight_context_internal"] != 0 else 0, vc, pc)):
            if wt != 0:
                loss_ref = loss_ref + wt * (contr(a, a, 0.2) + contr(b_, b_, 0.2)) / 2
        loss_ref.backward()
        got_in = [t.clone().requires_grad_(True) for t in sets]
        lcfg = cva.loss_fn.ContrastiveLossConfig(margin=0.2, **w)
        loss = cva.total_contrastive_loss(lcfg, *got_in, dtype="f32")
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref)), (float(loss), float(loss_ref))
        for k, (a, r) in enumerate(zip(got_in, ref_in)):
            assert float(r.grad.norm()) > 0, k  # (the data above violates some hinge terms of every set)
            err = float((a.grad.double() - r.grad).norm() / r.grad.norm())
            assert err <= 5e-6, (k, err)
    # the step API refuses the checker mode
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batch = cva.synthetic.make_batch(7, 6, [1, 2, 3, 4, 2, 1], 12, 10, 9, 6, dims[0], dims[1], ragged=False)
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.0)
    mgr.set_all_models_eval()
    for net in mgr.model_dict.values():
        net.set_compute_dtype("f32")
    tr = cva.RetrievalTrainer(cfg_x, mgr)
    with pytest.raises(RuntimeError, match="bf16 path only"):
        tr.train_step_native(batch)
