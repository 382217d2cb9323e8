"""Retrieval ranking on the MI355X (coot_retrieval_ranks) against the reference-generated golden ranks
(tests/golden/retrieval_metrics.npz, written by nntrainer/retrieval.py itself: oracle/gen_golden.py) and against the numpy
oracle (oracle/coot_oracle.py: compute_retrieval_cosine, the restatement of nntrainer/retrieval.py:68-98).

Integer work: the ranks are compared bit-exactly.  The only floating-point step is the similarity matrix; the kernel can hand
it out, so "the oracle's argsort on the matrix the kernel counted on" is an exact comparison, and the matrix itself is checked
against a float64 product (tolerance 2e-6 absolute on unit-norm rows: fp32 FMA chains of <= 768 terms)."""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _metrics_vec(res):
    return [res[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr", "sum")]


def test_golden_ranks_through_identity(env, golden_dir):
    """emb2 = identity makes the similarity matrix exactly the golden matrix d (products with 0 and 1 are exact), so the
    device ranks must equal the ranks nntrainer/retrieval.py produced for d and d^T.  Case 1 of the fixture has forced exact
    ties: there the reference's order is whatever numpy's introsort leaves, the kernel's rule is the reversal of a stable
    ascending sort — compared against that, and the golden ranks must agree wherever the row has no tie with its diagonal."""
    torch, cva = env
    from coot_videotext_amd.retrieval import retrieval_ranks_device
    g = np.load(os.path.join(golden_dir, "retrieval_metrics.npz"))
    for i in range(3):
        d = g[f"d{i}"].astype(np.float32)
        n = len(d)
        r12, r21, met, sim = retrieval_ranks_device(torch.from_numpy(d).cuda(), torch.eye(n, device="cuda"), normalize=False, want_sim=True)
        torch.cuda.synchronize()
        assert np.array_equal(sim.cpu().numpy(), d)
        stable = np.array([np.where(np.argsort(d[r], kind="stable")[::-1] == r)[0][0] for r in range(n)])
        stable_t = np.array([np.where(np.argsort(d.T[r], kind="stable")[::-1] == r)[0][0] for r in range(n)])
        assert np.array_equal(r12.cpu().numpy(), stable) and np.array_equal(r21.cpu().numpy(), stable_t)
        gold = g[f"ranks{i}"].astype(np.int64)
        tie_with_diag = np.array([(d[r] == d[r, r]).sum() > 1 for r in range(n)])
        assert np.array_equal(r12.cpu().numpy()[~tie_with_diag], gold[~tie_with_diag])
        if not tie_with_diag.any():  # metrics of the reference run itself (r1, r5, r10, r50, medr, meanr)
            assert np.allclose(met[0, :6].cpu().numpy(), g[f"res{i}"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,dim,normalize", [(700, 96, True), (333, 768, False), (64, 32, True), (1, 8, True), (2, 384, False)])
def test_ranks_and_metrics_match_oracle(env, n, dim, normalize):
    """Random embeddings with a planted match structure (so R@K is neither 0 nor 1), sizes that are not multiples of the
    64 x 64 tiles / 32-wide k chunks."""
    torch, cva = env
    from coot_videotext_amd.retrieval import retrieval_ranks_device, compute_retrieval_device
    rs = np.random.RandomState(n + dim)
    e1 = rs.randn(n, dim).astype(np.float32)
    e2 = (0.35 * e1 + rs.randn(n, dim)).astype(np.float32)
    if not normalize:
        e1 /= np.sqrt((e1 * e1).sum(-1, keepdims=True)); e2 /= np.sqrt((e2 * e2).sum(-1, keepdims=True))
    t1, t2 = torch.from_numpy(e1).cuda(), torch.from_numpy(e2).cuda()
    r12, r21, met, sim = retrieval_ranks_device(t1, t2, normalize=normalize, want_sim=True)
    torch.cuda.synchronize()
    sim = sim.cpu().numpy()
    # the similarity matrix: validate_epoch's normalisation (coot/trainer_retrieval.py:400-402) + emb1 . emb2^T
    a, b = e1.astype(np.float64), e2.astype(np.float64)
    if normalize:
        a = (e1 / np.sqrt((e1 * e1).sum(-1))[:, None]).astype(np.float64)
        b = (e2 / np.sqrt((e2 * e2).sum(-1))[:, None]).astype(np.float64)
    assert np.abs(sim - a @ b.T).max() < 2e-6
    # ranks and metrics: exact functions of that matrix
    res12, ranks12 = O.compute_retrieval_cosine(sim)
    res21, ranks21 = O.compute_retrieval_cosine(sim.T)
    assert np.array_equal(r12.cpu().numpy(), ranks12.astype(np.int64)) and np.array_equal(r21.cpu().numpy(), ranks21.astype(np.int64))
    assert np.allclose(met[0].cpu().numpy(), _metrics_vec(res12), rtol=1e-6, atol=1e-6)
    assert np.allclose(met[1].cpu().numpy(), _metrics_vec(res21), rtol=1e-6, atol=1e-6)
    # the reference-shaped front end, and R@K against the float64 truth within the path's tolerance (+-0.1 points)
    d12, d21, s1 = compute_retrieval_device(t1, t2, normalize=normalize)
    t12, _ = O.compute_retrieval_cosine(a @ b.T)
    for k in ("r1", "r5", "r10", "r50"):
        assert abs(100.0 * d12[k] - 100.0 * t12[k]) <= 0.1 + 100.0 / n * 0.5, (k, d12[k], t12[k])
    assert abs(s1 - (res12["r1"] + res21["r1"]) / 2) < 1e-6


def test_validate_epoch_uses_device_ranking(env):
    """RetrievalTrainer.validate_epoch (coot/trainer_retrieval.py:312-477, metric part) end to end on a few synthetic batches:
    the same dictionaries as the host oracle on the collected embeddings."""
    torch, cva = env
    from tests import helpers as H
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.05) for i in range(4)]
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0)
    tr = cva.RetrievalTrainer(cfg, mgr, is_test=True)
    batches = [cva.synthetic.make_batch(10 + i, 6, [1, 2, 3, 4, 2, 1], 12, 10, 9, 6, dims[0], dims[1], ragged=True) for i in range(3)]
    out = tr.validate_epoch(batches)
    # host recomputation from the same embeddings
    mgr.set_all_models_eval()
    vs, ps = [], []
    with torch.no_grad():
        for b in batches:
            vs.append(mgr.encode_visual(b).vid_emb.float().cpu().numpy())
            ps.append(mgr.encode_text(b).par_emb.float().cpu().numpy())
    v, p = np.concatenate(vs), np.concatenate(ps)
    v = v / np.sqrt((v * v).sum(-1))[:, None]; p = p / np.sqrt((p * p).sum(-1))[:, None]
    r12, r21, s1 = O.compute_retrieval(v.astype(np.float64), p.astype(np.float64))
    for k in ("r1", "r5", "r10", "r50", "medr"):
        assert abs(out["v2p"][k] - r12[k]) < 1e-6 and abs(out["p2v"][k] - r21[k]) < 1e-6, (k, out["v2p"], r12)
    assert abs(out["val_score_at_1"] - s1) < 1e-6 and np.isfinite(out["loss"])


def test_validate_epoch_embedding_export(env, tmp_path):
    """save_embs (coot/trainer_retrieval.py:404-415): the datasets of the reference's embeddings_<epoch>.h5 — keys, clip_num
    twice (sent_num is written from clip_num there), unit-norm rows next to the raw ones — and the file round trip."""
    torch, cva = env
    from tests import helpers as H
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.05) for i in range(4)]
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0)
    tr = cva.RetrievalTrainer(cfg, mgr, is_test=True)
    counts = [1, 2, 3, 4, 2, 1]
    batches = [cva.synthetic.make_batch(10 + i, 6, counts, 12, 10, 9, 6, dims[0], dims[1], ragged=True) for i in range(2)]
    for i, b in enumerate(batches):
        b.key = [f"vid{i}_{j}" for j in range(6)]
    out = tr.validate_epoch(batches, val_clips=False, save_embs=True, save_path=str(tmp_path / "embeddings_0.h5"))
    emb = out["embeddings"]
    names = ["vid_emb", "par_emb", "clip_emb", "sent_emb", "vid_context", "par_context"]
    assert set(emb) == {"clip_num", "sent_num", "key"} | set(names) | {n + "_before_norm" for n in names}
    assert emb["key"] == [f"vid{i}_{j}" for i in range(2) for j in range(6)]
    assert np.array_equal(emb["clip_num"], np.array(counts * 2)) and np.array_equal(emb["sent_num"], emb["clip_num"])
    mgr.set_all_models_eval()
    with torch.no_grad():
        v = [mgr.encode_visual(b) for b in batches]
        t = [mgr.encode_text(b) for b in batches]
    want = {"vid_emb": [x.vid_emb for x in v], "clip_emb": [x.clip_emb for x in v], "vid_context": [x.vid_context for x in v],
            "par_emb": [x.par_emb for x in t], "sent_emb": [x.sent_emb for x in t], "par_context": [x.par_context for x in t]}
    for n in names:
        raw = torch.cat(want[n], 0).float().cpu().numpy()
        assert emb[n + "_before_norm"].shape == raw.shape and np.array_equal(emb[n + "_before_norm"], raw), n
        assert np.abs(np.sqrt((emb[n].astype(np.float64) ** 2).sum(-1)) - 1).max() < 1e-6
        assert np.allclose(emb[n], raw / np.sqrt((raw * raw).sum(-1, keepdims=True)), rtol=1e-6, atol=1e-7)
    assert emb["clip_emb"].shape[0] == sum(counts) * 2 and emb["vid_emb"].shape == (12, 2 * dims[2])
    fn = out["embeddings_file"]
    assert os.path.exists(fn)
    if fn.endswith(".npz"):
        z = np.load(fn)
        assert set(z.files) == set(emb) and np.array_equal(z["vid_emb"], emb["vid_emb"]) and list(z["key"]) == emb["key"]
    else:
        import h5py
        with h5py.File(fn, "r") as h5:
            assert set(h5.keys()) == set(emb) and np.array_equal(h5["vid_emb"][()], emb["vid_emb"])
