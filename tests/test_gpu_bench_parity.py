"""Parity AT THE BENCHMARK SHAPES, default kernel dispatch, against fixtures the unmodified reference wrote
(oracle/gen_golden.py: gen_bench_*; coot/trainer_retrieval.py:253-291, coot/model_retrieval.py:86-197).

The token-tile chain kernels the benchmark spends its time in (infc_qkv_fwd, post_attn_fwd<8>, pre_attn_bwd, qkv_bwd, the
batched wide weight-gradient GEMM) only engage at >= 1024 tokens per call, i.e. at these sizes; nothing here touches
`fused_min_rows` or any other A/B switch, and every case asserts through coot_timing_collect that the fused family launched.

Two routes to the same fixture:
  * "autograd": RetrievalModelManager.encode_visual / encode_text + trainer loss hooks + loss.backward()
  * "native":   coot_step_forward (embeddings) and coot_train_step(do_optimizer = 0) (losses + every parameter gradient),
                the call bench.py times
Tolerances (north_star): embedding cosine >= 1 - 1e-3 per row; losses to 1e-3 relative; parameter gradients cosine > 0.99 on
the fixture's sub-sample and norm within 2 %.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

# bench_hbm_stress: BASELINE.json configs[4] per-GPU slice (Dv 1024, 64 clips per video: the global networks on Cmax = 64);
# bench_yc2_100m_2layer: configs[0] as BASELINE.json words it (2-layer local encoders at d_model 384 on the fused chains)
CASES = ["bench_anet", "bench_anet_ragged", "bench_yc2_100m", "bench_yc2_2d3d", "bench_hbm_stress", "bench_yc2_100m_2layer"]


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


_cache = {}


def _case(golden_dir, name):
    """Fixture + the seeded parameters / batch it was generated from (the generator stores only the seeds)."""
    if name not in _cache:
        _cache.clear()  # one ~0.6 GB host batch at a time
        g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
        seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
        cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph, layers=int(g["layers"]) if "layers" in g else 1)
        Ps = [O.make_params(cfgs[i], seed + 10 * i, scale=float(g["param_scale"])) for i in range(4)]
        b = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=bool(int(g["ragged"])), corr=0.5)
        _cache[name] = (g, cfgs, Ps, b)
    return _cache[name]


def _fused_launches(lib, cva):
    ms, fl, n = C.c_double(), C.c_double(), C.c_int()
    cva.lib.check(lib.coot_timing_collect(5, C.byref(ms), C.byref(fl), C.byref(n)), "timing_collect")
    return n.value


def _check_embeddings(g, got: dict, tag):
    worst = 1.0
    for key, val in got.items():
        cos = H.cosine_rows(val, g[key]).min()
        worst = min(worst, cos)
        print(f"[{tag}] {key}: min row cosine vs reference {cos:.6f}")
        assert cos > 1 - 1e-3, (key, cos)
    return worst


def _check_grads(g, grads_by_net: dict, tag):
    """grads_by_net[net][param name] -> ndarray.  Against gnorm:/gsub: of the fixture."""
    step = int(g["sub_step"])
    gmax = max(float(g[k]) for k in g if k.startswith("gnorm:"))
    bad, n_checked = [], 0
    for net, grads in grads_by_net.items():
        for pname, got in grads.items():
            key = f"{net}:{pname}"
            gn = float(g["gnorm:" + key])
            if gn < 1e-6 * gmax:  # mathematically zero gradient (softmax shift invariance): magnitude only
                if np.linalg.norm(got) > 1e-3 * gmax:
                    bad.append((key, "zero-grad", float(np.linalg.norm(got))))
                continue
            ref = g["gsub:" + key]
            c = H.cosine_flat(got.reshape(-1)[::(1 if ref.size == got.size else step)], ref)
            nr = float(np.linalg.norm(got.astype(np.float64))) / gn
            n_checked += 1
            if c < 0.995 or abs(nr - 1) > 0.02:
                print(f"[{tag}]   {key}: cos(sub) {c:.4f}  norm ratio {nr:.4f}  |g_ref| / max |g_ref| = {gn / gmax:.2e}")
            if not (c > 0.99 and 0.98 < nr < 1.02):
                bad.append((key, round(c, 4), round(nr, 4)))
    print(f"[{tag}] {n_checked} parameter gradients checked, {len(bad)} out of tolerance")
    assert not bad, bad
    assert n_checked >= 100  # 4 networks x (26 | 42) tensors, minus the zero-gradient ones
    return n_checked


@pytest.mark.parametrize("name", CASES)
def test_bench_shape_autograd_route(env, golden_dir, name):
    torch, cva = env
    lib = cva.lib.load()
    g, cfgs, Ps, b = _case(golden_dir, name)
    cfg, mgr = H.make_manager(cfgs, Ps, cc_weight=float(g["cc_weight"]))
    mgr.set_all_models_eval()
    trainer = cva.RetrievalTrainer(cfg, mgr, is_test=True)
    batch = cva.synthetic.batch_from_numpy(b)
    lib.coot_timing_enable(1)
    try:
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
        contr = trainer.compute_total_constrastive_loss(vis, txt)
        cc = trainer.compute_cyclecons_loss(vis, txt, torch.from_numpy(g["cc_idx_clip"]).cuda(), torch.from_numpy(g["cc_idx_sent"]).cuda())
        (contr + cc).backward()
        torch.cuda.synchronize()
        n_fused = _fused_launches(lib, cva)
    finally:
        lib.coot_timing_enable(0)
    assert n_fused >= 8, f"the fused token-tile kernels did not run ({n_fused} launches): this test must exercise the bench path"
    _check_embeddings(g, {"vid_emb": vis.vid_emb.detach().cpu().numpy(), "clip_emb": vis.clip_emb.detach().cpu().numpy(),
                          "vid_context": vis.vid_context.detach().cpu().numpy(), "par_emb": txt.par_emb.detach().cpu().numpy(),
                          "sent_emb": txt.sent_emb.detach().cpu().numpy(), "par_context": txt.par_context.detach().cpu().numpy()}, name)
    assert (vis.clip_emb_mask.cpu().numpy() == g["clip_emb_mask"]).all() and (vis.clip_emb_lens.cpu().numpy() == g["clip_emb_lens"]).all()
    assert (txt.sent_emb_mask.cpu().numpy() == g["sent_emb_mask"]).all() and (txt.sent_emb_lens.cpu().numpy() == g["sent_emb_lens"]).all()
    rc, rcc = float(g["contr_loss"]), float(g["cc_loss"])
    print(f"[{name}] contrastive {float(contr):.5f} vs {rc:.5f}; cycle-consistency {float(cc):.6f} vs {rcc:.6f}")
    assert abs(float(contr) - rc) < 1e-3 * abs(rc)
    assert abs(float(cc) - rcc) < 2e-3 * abs(rcc) + 1e-6
    _check_grads(g, {k: {n: p.grad.detach().cpu().numpy() for n, p in mgr.model_dict[k].named_parameters() if p.requires_grad}
                     for k in H.NET_KEYS}, name)
    # R@K of these embeddings against the reference's on the same inputs (north_star: +-0.1)
    for (a, c2, tag, a_key, c_key) in ((vis.vid_emb, txt.par_emb, "vp", "vid_emb", "par_emb"), (vis.clip_emb, txt.sent_emb, "cs", "clip_emb", "sent_emb")):
        e1 = torch.nn.functional.normalize(a.detach()).cpu().numpy()
        e2 = torch.nn.functional.normalize(c2.detach()).cpu().numpy()
        r12, r21, _ = cva.compute_retrieval(e1, e2)
        got = np.array([r12[k] for k in ("r1", "r5", "r10")] + [r21[k] for k in ("r1", "r5", "r10")])
        ref = g[f"ret_{tag}"][[0, 1, 2, 6, 7, 8]]
        # R@K are FRACTIONS; north_star's bound is +-0.1 percentage points — held on the N = 1 024 retrieval-parity set with a
        # trained state (test_gpu_rk_parity.py).  At N = 64 / 16 with seeded, UNTRAINED weights the similarities of a row differ in the
        # 4th decimal and one rank flip is 100 / N = 1.6 ... 6 pp, so the statement checked here is the one that bound stands for:
        # every comparison "item j scored above the paired item" that differs from the reference's is a near-tie of the REFERENCE's
        # own similarities, inside the embedding tolerance (1e-3 cosine) — no flip with a real margin.
        flips, margin = H.rank_flips(e1, e2, g[a_key], g[c_key])
        diff_pp = 100.0 * np.abs(got - ref).max()
        print(f"[{name}] R@1/5/10 {tag}: {got} vs reference {ref}  (max diff {diff_pp:.3f} pp at N = {e1.shape[0]}; "
              f"{flips} of {2 * e1.shape[0] * (e1.shape[0] - 1)} comparisons differ, largest reference margin among them {margin:.2e})")
        assert margin < 1e-3, (tag, flips, margin)
        assert flips <= 0.05 * 2 * e1.shape[0] * (e1.shape[0] - 1), (tag, flips)  # (untrained weights: a row's similarities differ in the 4th decimal)


@pytest.mark.parametrize("name", CASES)
def test_bench_shape_native_step(env, golden_dir, name):
    """coot_step_forward + coot_train_step(do_optimizer = 0): what bench.py times, dropout 0 so the fixture applies."""
    torch, cva = env
    lib = cva.lib.load()
    L = cva.lib
    g, cfgs, Ps, b = _case(golden_dir, name)
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=float(g["cc_weight"]))
    mgr.set_all_models_train()
    trainer = cva.RetrievalTrainer(cfg, mgr)
    batch = cva.synthetic.batch_from_numpy(b)
    idx = torch.from_numpy(np.concatenate([g["cc_idx_clip"], g["cc_idx_sent"]]).astype(np.int64)).cuda()
    lib.coot_timing_enable(1)
    try:
        losses = trainer.train_step_native(batch, do_optimizer=False, cc_indices=idx)
        torch.cuda.synchronize()
        n_fused = _fused_launches(lib, cva)
    finally:
        lib.coot_timing_enable(0)
    assert n_fused >= 8, f"the fused token-tile kernels did not run ({n_fused} launches)"
    total, contr, cc = (float(v) for v in losses)
    rc, rcc = float(g["contr_loss"]), float(g["cc_loss"])
    print(f"[{name} native] contrastive {contr:.5f} vs {rc:.5f}; cycle-consistency {cc:.6f} vs {rcc:.6f}")
    assert abs(contr - rc) < 1e-3 * abs(rc) and abs(cc - rcc) < 2e-3 * abs(rcc) + 1e-6 and abs(total - contr - cc) < 1e-5
    grads = {}
    for k in H.NET_KEYS:
        net = mgr.model_dict[k]
        flat = net._grad_flat.detach().cpu().numpy()
        grads[k] = {n: flat[off:off + int(np.prod(shape))].reshape(shape) for (n, off, shape) in net.table}
    _check_grads(g, grads, name + " native")
    # the embeddings of the same native forward (coot_step_forward: the data-parallel phase entry point, same side_forward)
    st = trainer._native
    B, Nc, D = st.dims.B, st.dims.Nc, 384
    dev = batch.vid_feat.device
    local_v, local_t = torch.empty(B + Nc, D, device=dev), torch.empty(B + Nc, D, device=dev)
    glob_v, glob_t = torch.empty(B, 2 * D, device=dev), torch.empty(B, 2 * D, device=dev)
    resh_v, resh_t = torch.empty(B, st.dims.Cmax_clip, D, device=dev), torch.empty(B, st.dims.Cmax_sent, D, device=dev)
    _, x = trainer._native_setup(batch)
    main = torch.cuda.current_stream()
    L.check(lib.coot_step_forward(C.byref(st.cfg), C.byref(st.bufs), C.byref(x), C.byref(st.dims), local_v.data_ptr(), local_t.data_ptr(),
                                  glob_v.data_ptr(), glob_t.data_ptr(), resh_v.data_ptr(), resh_t.data_ptr(), st.ws.data_ptr(), st.ws.numel(),
                                  1, 0, 0, main.cuda_stream, main.cuda_stream, st.streams[1].cuda_stream), "coot_step_forward")
    torch.cuda.synchronize()
    lv, lt = local_v.cpu().numpy(), local_t.cpu().numpy()
    _check_embeddings(g, {"vid_emb": glob_v.cpu().numpy(), "par_emb": glob_t.cpu().numpy(), "vid_context": lv[:B], "clip_emb": lv[B:],
                          "par_context": lt[:B], "sent_emb": lt[B:]}, name + " native")
    # packed [B, Cmax, D] copies: exactly the flat rows scattered by count (coot/model_retrieval.py:121-136)
    resh_ref, _ = O.pack_by_count(lv[B:], b["clip_num"])[:2]
    assert np.array_equal(resh_v.cpu().numpy(), resh_ref.astype(np.float32))
