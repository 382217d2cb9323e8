"""Seeded synthetic RetrievalDataBatchTuple generators (ActivityNet / YouCook2 shaped), SURVEY 8d.
Features ~ N(0,1), zero in padded rows, masks/lengths consistent; generated directly on the device."""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from .model_retrieval import RetrievalDataBatchTuple, packed_index

# fixed-shape workloads of BASELINE.json / SURVEY 8d: (B, C, Lv, Lc, Lp, Ls, Dv, Dt)
WORKLOADS = {
    "anet": dict(B=64, C=4, Lv=80, Lc=80, Lp=64, Ls=16, Dv=2048, Dt=1536),
    "yc2_100m": dict(B=16, C=8, Lv=80, Lc=20, Lp=96, Ls=12, Dv=512, Dt=1536),
    "yc2_2d3d": dict(B=64, C=8, Lv=80, Lc=20, Lp=96, Ls=12, Dv=4096, Dt=1536),
    # BASELINE.json configs[3] words the concatenated 2D + 3D features as d = 2816 (the shipped YAML says 4096): same network, other width
    "yc2_2d3d_2816": dict(B=64, C=8, Lv=80, Lc=20, Lp=96, Ls=12, Dv=2816, Dt=1536),
    # BASELINE.json configs[4], per-GPU slice: 64 clips x 80 frames x d = 1024 per video ("HBM-bandwidth roofline stress": the input
    # stream Nc Lc Dv is what matters; 100k such videos are streamed batch by batch, a batch is what one step sees)
    "hbm_stress": dict(B=16, C=64, Lv=80, Lc=80, Lp=64, Ls=16, Dv=1024, Dt=1536),
}
# ragged ActivityNet-shaped workload (SURVEY 8d "ragged variant"): clips per video ~ the annotation statistics (mean 3.7, max 27),
# frames per clip / video uniform in [L / 8, L] resp. [L / 4, L], words uniform up to Ls = 30 — what a real batch looks like; the
# reference pads it to the batch maxima.  C = 0 marks "drawn per video" (anet_like_counts).
WORKLOADS["anet_ragged"] = dict(B=64, C=0, Lv=80, Lc=80, Lp=64, Ls=30, Dv=2048, Dt=1536)


def anet_like_counts(seed: int, B: int) -> np.ndarray:
    """Clips per video with the shape of the ActivityNet annotation statistics (mean 3.74, p95 7, max 27)."""
    rs = np.random.RandomState(seed)
    return np.minimum(27, 1 + rs.negative_binomial(2, 0.42, B)).astype(np.int64)


# workload -> (YAML in config/retrieval, overrides of dataset_*.vid_feat_dim)
WORKLOAD_CONFIG = {
    "anet": ("anet_coot", None), "yc2_100m": ("yc2_100m_coot", None), "yc2_2d3d": ("yc2_2d3d_coot", None),
    "yc2_2d3d_2816": ("yc2_2d3d_coot", 2816), "hbm_stress": ("anet_coot", 1024), "anet_ragged": ("anet_coot", None),
}
WORKLOAD_LABEL = {
    "anet": "ActivityNet-shaped paper config (anet_coot)", "yc2_100m": "YouCook2 (100M features)-shaped paper config (yc2_100m_coot)",
    "yc2_2d3d": "YouCook2 (2D+3D features)-shaped paper config (yc2_2d3d_coot)",
    "yc2_2d3d_2816": "YouCook2 (2D+3D features, d = 2816 as BASELINE.json words it)-shaped config (yc2_2d3d_coot, vid_feat_dim 2816)",
    "hbm_stress": "HBM-stress slice of BASELINE.json configs[4] (anet_coot networks, vid_feat_dim 1024)",
    "anet_ragged": "ActivityNet-shaped ragged batch (anet_coot; clip counts ~ annotation statistics, ragged frame / word counts)",
}


def make_batch(seed: int, B: int, counts: Union[int, Sequence[int]], Lv: int, Lc: int, Lp: int, Ls: int, Dv: int, Dt: int,
               ragged: bool = False, device="cuda", packed: bool = False) -> RetrievalDataBatchTuple:
    g = torch.Generator(device="cpu").manual_seed(seed)
    counts_t = torch.full((B,), int(counts), dtype=torch.long) if np.isscalar(counts) else torch.as_tensor(counts, dtype=torch.long)
    Nc = int(counts_t.sum())

    def lens(n, L, lo):
        if not ragged:
            return torch.full((n,), L, dtype=torch.long)
        l = torch.randint(lo, L + 1, (n,), generator=g)
        l[int(torch.randint(0, n, (1,), generator=g))] = L
        return l

    dg = torch.Generator(device=device).manual_seed(seed + 1)

    def feats(n, L, D, ln):
        x = torch.randn(n, L, D, device=device, generator=dg)
        mask = torch.arange(L, device=device)[None, :] >= ln.to(device)[:, None]
        x.masked_fill_(mask[:, :, None], 0.0)
        return x, mask

    vl, pl = lens(B, Lv, max(1, Lv // 4)), lens(B, Lp, max(1, Lp // 4))
    cl, sl = lens(Nc, Lc, max(1, Lc // 8)), lens(Nc, Ls, max(1, Ls // 4))
    vf, vm = feats(B, Lv, Dv, vl)
    pf, pm = feats(B, Lp, Dt, pl)
    cf, cm = feats(Nc, Lc, Dv, cl)
    sf, sm = feats(Nc, Ls, Dt, sl)
    keys = [str(i) for i in range(B)]
    batch = RetrievalDataBatchTuple(
        key=keys, data_key=keys, sentences=[[""]] * B, vid_feat=vf, vid_feat_mask=vm, vid_feat_len=vl.to(device),
        par_feat=pf, par_feat_mask=pm, par_feat_len=pl.to(device), clip_num=counts_t.to(device), clip_feat=cf,
        clip_feat_mask=cm, clip_feat_len=cl.to(device), sent_num=counts_t.to(device), sent_feat=sf, sent_feat_mask=sm,
        sent_feat_len=sl.to(device), max_clip_num=int(counts_t.max()), max_sent_num=int(counts_t.max()))
    if packed:  # row starts of the valid tokens, from the host-side lengths (no device sync)
        batch.cu_vis, batch.tok_vis = packed_index(vl, cl)
        batch.cu_txt, batch.tok_txt = packed_index(pl, sl)
        batch.cu_vis, batch.cu_txt = batch.cu_vis.to(device), batch.cu_txt.to(device)
    return batch


def batch_from_numpy(b: dict, device="cuda", packed: bool = False) -> RetrievalDataBatchTuple:
    """dict of numpy arrays (same field names as the batch tuple) -> device batch."""
    t = {k: torch.as_tensor(np.asarray(v)) for k, v in b.items()}
    B = len(b["clip_num"])
    keys = [str(i) for i in range(B)]
    f = lambda k: t[k].float().to(device)
    batch = RetrievalDataBatchTuple(
        key=keys, data_key=keys, sentences=[[""]] * B, vid_feat=f("vid_feat"), vid_feat_mask=t["vid_feat_mask"].to(device),
        vid_feat_len=t["vid_feat_len"].to(device), par_feat=f("par_feat"), par_feat_mask=t["par_feat_mask"].to(device),
        par_feat_len=t["par_feat_len"].to(device), clip_num=t["clip_num"].to(device), clip_feat=f("clip_feat"),
        clip_feat_mask=t["clip_feat_mask"].to(device), clip_feat_len=t["clip_feat_len"].to(device),
        sent_num=t["sent_num"].to(device), sent_feat=f("sent_feat"), sent_feat_mask=t["sent_feat_mask"].to(device),
        sent_feat_len=t["sent_feat_len"].to(device), max_clip_num=int(np.max(b["clip_num"])),
        max_sent_num=int(np.max(b["sent_num"])))
    if packed:
        batch.cu_vis, batch.tok_vis = packed_index(t["vid_feat_len"], t["clip_feat_len"])
        batch.cu_txt, batch.tok_txt = packed_index(t["par_feat_len"], t["sent_feat_len"])
        batch.cu_vis, batch.cu_txt = batch.cu_vis.to(device), batch.cu_txt.to(device)
    return batch
