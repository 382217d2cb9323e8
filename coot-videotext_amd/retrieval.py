"""Retrieval metrics, nntrainer/retrieval.py:31-98 semantics: rank of the diagonal item in argsort(row)[::-1]; R@K as
fractions; medr = floor(median)+1; meanr = mean+1.

compute_retrieval / compute_retrieval_cosine: host (numpy) mirror of the reference functions, for CPU tensors.
compute_retrieval_device: the same results for embeddings that live on the MI355X, computed by libcoot_hip.so
(coot_retrieval_ranks: normalisation, similarities, both rank vectors and the metric dictionaries in four launches, the
N x N matrix is never materialised).  No CPU fallback: CUDA tensors in, device kernels or an error."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

VALKEYS = ["r1", "r5", "r10", "r50", "medr", "meanr", "sum"]  # nntrainer/retrieval.py:12


def compute_retrieval_cosine(dot_product: np.ndarray) -> Tuple[Dict[str, float], np.ndarray, np.ndarray]:
    n = len(dot_product)
    order = np.argsort(dot_product, axis=1)[:, ::-1]
    ranks = np.argmax(order == np.arange(n)[:, None], axis=1).astype(np.float64)
    top1 = order[:, 0].astype(np.float64)
    r1, r5, r10, r50 = [float((ranks < k).mean()) for k in (1, 5, 10, 50)]
    medr = float(np.floor(np.median(ranks)) + 1)
    meanr = float(ranks.mean() + 1)
    return {"r1": r1, "r5": r5, "r10": r10, "r50": r50, "medr": medr, "meanr": meanr, "sum": r1 + r5 + r50}, top1, ranks


def compute_retrieval(emb1: np.ndarray, emb2: np.ndarray):
    d = np.dot(emb1, emb2.T)
    res1, _, _ = compute_retrieval_cosine(d)
    res2, _, _ = compute_retrieval_cosine(d.T)
    return res1, res2, (res1["r1"] + res2["r1"]) / 2


def retrieval_ranks_device(emb1, emb2, normalize: bool = False, want_sim: bool = False):
    """emb1, emb2: cuda float32 [N, d].  Returns (ranks_12 int32 [N], ranks_21 int32 [N], metrics float32 [2, 7], sim or None),
    all on the device (no synchronisation)."""
    import torch
    from . import lib as _lib
    if not (emb1.is_cuda and emb2.is_cuda):
        raise RuntimeError("retrieval_ranks_device needs CUDA tensors (there is no CPU fallback; use compute_retrieval)")
    assert emb1.dtype == torch.float32 and emb2.dtype == torch.float32 and emb1.shape == emb2.shape and emb1.dim() == 2
    emb1, emb2 = emb1.contiguous(), emb2.contiguous()
    n, d = emb1.shape
    lib = _lib.load()
    ws = torch.empty(lib.coot_retrieval_workspace_bytes(n, d), dtype=torch.uint8, device=emb1.device)
    r12 = torch.empty(n, dtype=torch.int32, device=emb1.device)
    r21 = torch.empty(n, dtype=torch.int32, device=emb1.device)
    met = torch.empty(2, 7, dtype=torch.float32, device=emb1.device)
    sim = torch.empty(n, n, dtype=torch.float32, device=emb1.device) if want_sim else None
    _lib.check(lib.coot_retrieval_ranks(emb1.data_ptr(), emb2.data_ptr(), n, d, int(normalize), r12.data_ptr(), r21.data_ptr(),
                                        met.data_ptr(), sim.data_ptr() if want_sim else None, ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream().cuda_stream), "coot_retrieval_ranks")
    return r12, r21, met, sim


def compute_retrieval_device(emb1, emb2, normalize: bool = False):
    """Device version of compute_retrieval (nntrainer/retrieval.py:31-65): (res_1to2, res_2to1, sum_at_1) with the
    reference's dictionary keys.  One 56-byte D2H copy."""
    _, _, met, _ = retrieval_ranks_device(emb1, emb2, normalize)
    m = met.cpu().numpy().astype(np.float64)
    res1 = {k: float(v) for k, v in zip(VALKEYS, m[0])}
    res2 = {k: float(v) for k, v in zip(VALKEYS, m[1])}
    return res1, res2, (res1["r1"] + res2["r1"]) / 2
