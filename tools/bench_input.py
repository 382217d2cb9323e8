"""PCIe-inclusive rate of the train step (input row): host data points -> collate_fn into a pinned arena
(coot_collate_level) -> one async H2D copy per batch on the copy stream (DeviceLoader) -> coot_train_step, on the
ActivityNet-shaped batch (64 videos x 4 clips x 80 frames x 2048, 260 MB fp32 per batch).  Prints one JSON line per setting.

    python tools/bench_input.py [--batches 24] [--threads 4 16] [--bf16 0 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coot_videotext_amd as cva  # noqa: E402
from coot_videotext_amd.dataset_retrieval import DeviceLoader, RetrievalDataPointTuple, collate_fn, BatchArena  # noqa: E402


def make_points(seed, B=64, C=4, L=80, Ls=16, dv=2048, dt=1536):
    rs = np.random.default_rng(seed)
    pts = []
    for b in range(B):
        clips = [rs.standard_normal((L, dv), dtype=np.float32) for _ in range(C)]
        par = rs.standard_normal((C * Ls, dt), dtype=np.float32)
        sents = [par[i * Ls:(i + 1) * Ls] for i in range(C)]
        pts.append(RetrievalDataPointTuple(f"v{seed}_{b}", f"v{seed}_{b}", ["w"] * C, rs.standard_normal((L, dv), dtype=np.float32), L, par,
                                           C * Ls, C, clips, [L] * C, C, sents, [Ls] * C))
    return pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--threads", type=int, nargs="+", default=[4, 16])
    ap.add_argument("--bf16", type=int, nargs="+", default=[0, 1])
    a = ap.parse_args()
    cfg = cva.load_named_config("anet_coot")
    mgr = cva.RetrievalModelManager(cfg).cuda()
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg, mgr)
    distinct = [make_points(s) for s in range(2)]
    pairs = sum(p.clip_num for p in distinct[0])
    # host-only collation rate
    for th in a.threads:
        ar = BatchArena(pin=True)
        collate_fn(distinct[0], ar, threads=th)
        t0 = time.perf_counter()
        for i in range(6):
            collate_fn(distinct[i % 2], ar, threads=th)
        dt = (time.perf_counter() - t0) / 6
        print(json.dumps({"stage": "collate only", "threads": th, "ms_per_batch": round(dt * 1e3, 2), "GB_per_s": round(ar.nbytes / dt / 1e9, 2)}))
    for bf16 in a.bf16:
        for th in a.threads:
            src = [distinct[i % 2] for i in range(a.batches + 4)]
            loader = DeviceLoader(src, depth=2, bf16=bool(bf16), threads=th)
            t0 = None
            for i, batch in enumerate(loader):
                if i == 4:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                tr.train_step_native(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.batches
            print(json.dumps({"stage": "collate + H2D + train step", "bf16_staging": bool(bf16), "threads": th, "ms_per_step": round(dt * 1e3, 3),
                              "clip_pairs_per_s": round(pairs / dt, 1), "arena_MB": round(loader.host[0].nbytes / 1e6, 1)}))


if __name__ == "__main__":
    main()
