"""Retrieval-metric parity on the SURVEY 8d parity set: 1 024 videos / ~3 900 clips generated from shared latent codes, a
'trained-like' state (the reference trained for 160 Adam steps on that distribution, int8-quantised so the fixture stays small;
oracle/gen_golden.py: gen_rk_parity).  The fixture holds R@1/5/10/50, MedR, MeanR of the REFERENCE's fp32 CPU embeddings through
nntrainer/retrieval.py for video<->paragraph and clip<->sentence, both directions.  Here: the same state and inputs through
validate_epoch (HIP forward in bf16, device ranking) — north_star: R@K within +-0.1 (percentage points).
Matches coot/trainer_retrieval.py:312-477, nntrainer/retrieval.py:68-98."""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_rk_parity_set(golden_dir):
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    g = np.load(os.path.join(golden_dir, "rk_parity.npz"))
    N, Bsz, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph, _steps = [int(v) for v in g["meta"]]
    cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph)
    Ps = []
    for k in H.NET_KEYS:
        P = {}
        for name in g.files:
            if name.startswith(f"q:{k}:"):
                n = name[len(f"q:{k}:"):]
                P[n] = g[name].astype(np.float32) * np.float32(g[f"s:{k}:{n}"])
        Ps.append(P)
    assert [len(P) for P in Ps] == [25, 34, 25, 34], [len(P) for P in Ps]  # every state-dict entry but the positional table
    cfg, mgr = H.make_manager(cfgs, Ps)
    mgr.set_all_models_eval()
    tr = cva.RetrievalTrainer(cfg, mgr, is_test=True)
    batches = []
    for i in range(N // Bsz):
        seed = 900000 + 13 * i
        noise, clusters, spread = float(g["eval_gen"][0]), int(g["eval_gen"][1]), float(g["eval_gen"][2])
        b = O.make_latent_batch(seed, Bsz, O.anet_like_counts(seed + 1, Bsz), Lv, Lc, Lp, Ls, dv, dt, noise=noise, clusters=clusters, spread=spread)
        batches.append(cva.synthetic.batch_from_numpy(b))
    out = tr.validate_epoch(batches, val_clips=True, save_embs=True)
    emb = out["embeddings"]
    assert emb["clip_emb"].shape[0] == int(g["n_clips"]) and emb["vid_emb"].shape[0] == N
    # a few rows of the reference's embeddings (every 37th): cosine per row
    for k in ("vid_emb", "par_emb", "clip_emb", "sent_emb"):
        cos = H.cosine_rows(emb[k + "_before_norm"][::37], g["rows:" + k]).min()
        print(f"[rk parity] {k}: min row cosine vs reference {cos:.6f}")
        assert cos > 1 - 1e-3, (k, cos)
    keys = ("r1", "r5", "r10", "r50", "medr", "meanr")
    worst = 0.0
    for tag, a, b_ in (("vp", "v2p", "p2v"), ("cs", "c2s", "s2c")):
        ref = g["ret_" + tag]
        for d, (name, off) in enumerate(((a, 0), (b_, 6))):
            got = np.array([float(out[name][k]) for k in keys])
            want = ref[off:off + 6]
            print(f"[rk parity] {name}: R@1/5/10/50 {100 * got[:4]} vs reference {100 * want[:4]}; MedR {got[4]} vs {want[4]}; "
                  f"MeanR {got[5]:.2f} vs {want[5]:.2f}")
            worst = max(worst, float(np.abs(100 * got[:4] - 100 * want[:4]).max()))
            assert np.abs(100 * got[:4] - 100 * want[:4]).max() <= 0.1 + 1e-9, (name, got, want)   # +-0.1 percentage points
            assert abs(got[4] - want[4]) <= 1 and abs(got[5] - want[5]) <= 0.01 * want[5] + 0.05, (name, got, want)
    assert 0.2 < float(g["ret_cs"][0]) < 0.95, "the parity set must retrieve far above chance and below saturation, or the comparison is vacuous"
    print(f"[rk parity] largest R@K deviation {worst:.3f} percentage points")
