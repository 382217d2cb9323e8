"""Packed (variable-length) token rows of the local networks, SURVEY 8f-2 (the reference pads every sequence to the batch
maximum, coot/dataset_retrieval.py:335-463, and runs the padding through the networks): with cu_seqlens the valid tokens only.

  * packed vs padded on the same ragged batch: pooled embeddings, losses and every parameter gradient agree (padded rows carry
    exactly zero pooling weight and never reach a valid row: poolers.py:190, transformer_legacy.py:544) — network level, the
    autograd route and the native step;
  * the packed path against the reference-generated ragged fixture (tests/golden/bench_anet_ragged.npz), i.e. the same
    tolerances the padded path is held to in test_gpu_bench_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H
from tests.test_gpu_bench_parity import _case, _check_embeddings, _check_grads, _fused_launches

pytestmark = pytest.mark.gpu

ANET_LOCAL = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
TEXT_LOCAL = O.NetConfig(input_dim=1536, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


@pytest.mark.parametrize("name,cfg,N1,L1,N2,L2,train", [("video", ANET_LOCAL, 9, 80, 40, 37, False), ("text", TEXT_LOCAL, 16, 64, 90, 16, False),
                                                        ("video_train", ANET_LOCAL, 7, 64, 33, 80, True), ("min_len", TEXT_LOCAL, 160, 20, 3, 128, False)])
def test_local_network_packed_rows_match_padded(env, name, cfg, N1, L1, N2, L2, train):
    torch, cva = env
    rs = np.random.RandomState(len(name) + N1)
    P = O.make_params(cfg, 51)

    def seqs(n, L):
        lens = rs.randint(1, L + 1, size=n)
        lens[0] = L
        if name == "min_len":
            lens[1:3] = 1
        x = rs.randn(n, L, cfg.input_dim)
        x[np.arange(L)[None, :] >= lens[:, None]] = 0
        return x, lens

    (x1, l1), (x2, l2) = seqs(N1, L1), seqs(N2, L2)
    R = rs.randn(N1 + N2, cfg.hidden_dim)
    assert l1.sum() + l2.sum() >= 1024, "below 1024 tokens the packed path is not taken"
    res = []
    lib = cva.lib.load()
    for packed in (False, True):
        net = H.make_hip_net(cfg, P, dropout=0.1 if train else 0.0)
        net.train(train)
        t1, t2 = torch.from_numpy(x1).float().cuda(), torch.from_numpy(x2).float().cuda()
        n1, n2 = torch.from_numpy(l1).cuda(), torch.from_numpy(l2).cuda()
        pk = cva.packed_index(torch.from_numpy(l1), torch.from_numpy(l2)) if packed else None
        if pk is not None:
            pk = (pk[0].cuda(), pk[1])
            assert pk[1] == int(l1.sum() + l2.sum()) and pk[1] < N1 * L1 + N2 * L2
        lib.coot_timing_enable(1)
        a, b = net.forward_pair(t1, n1, t2, n2, seed=99, packed=pk)
        torch.cuda.synchronize()
        ms, by, nl = C.c_double(), C.c_double(), C.c_int()
        cva.lib.check(lib.coot_timing_collect(7, C.byref(ms), C.byref(by), C.byref(nl)), "timing_collect")
        lib.coot_timing_enable(0)
        rows_processed = int(round(by.value / (6.0 * cfg.input_dim)))  # the input LayerNorm reports 6 bytes per element it touches
        assert rows_processed == (int(l1.sum() + l2.sum()) if packed else N1 * L1 + N2 * L2), (packed, rows_processed)
        pooled = torch.cat([a, b], 0)
        (pooled * torch.from_numpy(R).float().cuda()).sum().backward()
        torch.cuda.synchronize()
        res.append((pooled.detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters() if p.requires_grad}))
    (p0, g0), (p1, g1) = res
    if train:  # different row numbering -> different dropout masks: the two runs agree statistically only
        cos = H.cosine_rows(p1, p0).mean()
        print(f"[{name}] packed vs padded (dropout 0.1, different masks): mean row cosine {cos:.4f}")
        assert cos > 0.7 and np.isfinite(p1).all() and all(np.isfinite(v).all() for v in g1.values())
        return
    ep = H.rel_err(p1, p0)
    print(f"[{name}] packed vs padded: pooled rel err {ep:.2e}")
    assert ep < 2e-3
    bad, table = H.grad_report([(n, g) for n, g in g1.items() if "key_projection.bias" not in n], g0, cos_min=0.999, ratio_tol=0.01)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("route", ["autograd", "native"])
def test_packed_rows_against_the_reference_ragged_fixture(env, golden_dir, route):
    """bench_anet_ragged (64 videos, ANet-like clip counts, lengths up to 80 / 30) through the PACKED path against what the
    unmodified reference computed on the padded batch."""
    torch, cva = env
    lib = cva.lib.load()
    g, cfgs, Ps, b = _case(golden_dir, "bench_anet_ragged")
    batch = cva.synthetic.batch_from_numpy(b, packed=True)
    pad_v = batch.vid_feat.shape[0] * batch.vid_feat.shape[1] + batch.clip_feat.shape[0] * batch.clip_feat.shape[1]
    pad_t = batch.par_feat.shape[0] * batch.par_feat.shape[1] + batch.sent_feat.shape[0] * batch.sent_feat.shape[1]
    print(f"[varlen] valid / padded tokens: video side {batch.tok_vis} / {pad_v}, text side {batch.tok_txt} / {pad_t}")
    assert 1024 <= batch.tok_vis < 0.8 * pad_v and 1024 <= batch.tok_txt < 0.8 * pad_t
    idx_c, idx_s = torch.from_numpy(g["cc_idx_clip"]).cuda(), torch.from_numpy(g["cc_idx_sent"]).cuda()
    if route == "autograd":
        cfg, mgr = H.make_manager(cfgs, Ps, cc_weight=float(g["cc_weight"]))
        mgr.set_all_models_eval()
        trainer = cva.RetrievalTrainer(cfg, mgr, is_test=True)
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
        contr = trainer.compute_total_constrastive_loss(vis, txt)
        cc = trainer.compute_cyclecons_loss(vis, txt, idx_c, idx_s)
        (contr + cc).backward()
        torch.cuda.synchronize()
        _check_embeddings(g, {"vid_emb": vis.vid_emb.detach().cpu().numpy(), "clip_emb": vis.clip_emb.detach().cpu().numpy(),
                              "vid_context": vis.vid_context.detach().cpu().numpy(), "par_emb": txt.par_emb.detach().cpu().numpy(),
                              "sent_emb": txt.sent_emb.detach().cpu().numpy(), "par_context": txt.par_context.detach().cpu().numpy()}, "packed")
        contr, cc = float(contr), float(cc)
        grads = {k: {n: p.grad.detach().cpu().numpy() for n, p in mgr.model_dict[k].named_parameters() if p.requires_grad} for k in H.NET_KEYS}
    else:
        cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=float(g["cc_weight"]))
        mgr.set_all_models_train()
        trainer = cva.RetrievalTrainer(cfg, mgr)
        losses = trainer.train_step_native(batch, do_optimizer=False, cc_indices=torch.cat([idx_c, idx_s]))
        torch.cuda.synchronize()
        assert trainer._native.dims.tok_vis == batch.tok_vis and trainer._native.dims.tok_txt == batch.tok_txt
        _, contr, cc = (float(v) for v in losses)
        grads = {}
        for k in H.NET_KEYS:
            net = mgr.model_dict[k]
            flat = net._grad_flat.detach().cpu().numpy()
            grads[k] = {n: flat[off:off + int(np.prod(shape))].reshape(shape) for (n, off, shape) in net.table}
    rc, rcc = float(g["contr_loss"]), float(g["cc_loss"])
    print(f"[varlen {route}] contrastive {contr:.5f} vs {rc:.5f}; cycle-consistency {cc:.6f} vs {rcc:.6f}")
    assert abs(contr - rc) < 1e-3 * abs(rc) and abs(cc - rcc) < 2e-3 * abs(rcc) + 1e-6
    _check_grads(g, grads, f"packed {route}")
