"""Retrieval metrics (host side), nntrainer/retrieval.py:31-98 semantics: rank of the diagonal item in
argsort(row)[::-1]; R@K as fractions; medr = floor(median)+1; meanr = mean+1."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


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
