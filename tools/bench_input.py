"""PCIe-inclusive rate of the train step (input row): host data points -> collate_fn into a pinned arena
(coot_collate_level) -> one async H2D copy per batch on the copy stream (DeviceLoader) -> coot_train_step, on the
ActivityNet-shaped batch (64 videos x 4 clips x 80 frames x 2048, 260 MB fp32 per batch).  Prints one JSON line per setting.

    python tools/bench_input.py [--batches 24] [--threads 4 16] [--bf16 0 1] [--ragged] [--packed 0 1]

--ragged: ActivityNet-like ragged batches (clips per video ~ the annotation statistics, frames per clip uniform in [10, 80], per video in
[20, 80], words per sentence in [4, 30]: synthetic.WORKLOADS["anet_ragged"]); --packed 1: collated PACKED AT THE SOURCE
(coot_collate_packed: no padding row is written, copied over PCIe or read from HBM; bf16 rows are consumed as they are).
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


def make_points_ragged(seed, B=64, L=80, Ls=30, dv=2048, dt=1536):
    rs = np.random.default_rng(seed)
    counts = cva.synthetic.anet_like_counts(4321 + seed, B)
    pts = []
    for b in range(B):
        Cn = int(counts[b])
        clips = [rs.standard_normal((int(rs.integers(L // 8, L + 1)), dv), dtype=np.float32) for _ in range(Cn)]
        sl = [int(rs.integers(4, Ls + 1)) for _ in range(Cn)]
        while sum(sl) > 128:  # paragraphs of <= 128 words (the packed-row attention kernels' limit; ANet paragraphs average ~50)
            sl = [max(3, n - 1) for n in sl]
        par = rs.standard_normal((sum(sl), dt), dtype=np.float32)
        sents, ptr = [], 0
        for n in sl:
            sents.append(par[ptr:ptr + n]); ptr += n
        Lv = int(rs.integers(L // 4, L + 1))
        pts.append(RetrievalDataPointTuple(f"v{seed}_{b}", f"v{seed}_{b}", ["w"] * Cn, rs.standard_normal((Lv, dv), dtype=np.float32), Lv, par,
                                           par.shape[0], Cn, clips, [c.shape[0] for c in clips], Cn, sents, sl))
    return pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--packed", type=int, nargs="+", default=[0])
    ap.add_argument("--background", type=int, nargs="+", default=[0], help="1: collation + H2D enqueue on a loader thread")
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--threads", type=int, nargs="+", default=[4, 16])
    ap.add_argument("--bf16", type=int, nargs="+", default=[0, 1])
    a = ap.parse_args()
    cfg = cva.load_named_config("anet_coot")
    mgr = cva.RetrievalModelManager(cfg).cuda()
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg, mgr)
    distinct = [(make_points_ragged if a.ragged else make_points)(s) for s in range(2)]
    pairs = sum(p.clip_num for p in distinct[0])
    if a.ragged:  # resident baseline of the same ragged batch (packed rows inside the device, features resident in HBM)
        hb = collate_fn(distinct[0])
        hb.to_cuda()
        cva.attach_packed_index(hb)
        for _ in range(5):
            tr.train_step_native(hb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            tr.train_step_native(hb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(json.dumps({"stage": "train step, features resident in HBM (padded tensors + cu_seqlens)", "ms_per_step": round(dt * 1e3, 3),
                          "clip_pairs_per_s": round(pairs / dt, 1), "clip_pairs": int(pairs)}))
    # host-only collation rate
    for pk in a.packed:
      for th in a.threads:
        ar = BatchArena(pin=True)
        collate_fn(distinct[0], ar, threads=th, packed=bool(pk))
        t0 = time.perf_counter()
        for i in range(6):
            collate_fn(distinct[i % 2], ar, threads=th, packed=bool(pk))
        dt = (time.perf_counter() - t0) / 6
        print(json.dumps({"stage": "collate only", "packed_at_source": bool(pk), "threads": th, "ms_per_batch": round(dt * 1e3, 2),
                          "arena_MB": round(ar.nbytes / 1e6, 1), "GB_per_s": round(ar.nbytes / dt / 1e9, 2)}))
    for bg in a.background:
     for pk in a.packed:
      for bf16 in a.bf16:
        for th in a.threads:
            src = [distinct[i % 2] for i in range(a.batches + 4)]
            loader = DeviceLoader(src, depth=2, bf16=bool(bf16), threads=th, packed=bool(pk), background=bool(bg))
            t0 = None
            for i, batch in enumerate(loader):
                if i == 4:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                tr.train_step_native(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.batches
            print(json.dumps({"stage": "collate + H2D + train step", "loader_thread": bool(bg), "packed_at_source": bool(pk), "bf16_staging": bool(bf16), "threads": th, "ms_per_step": round(dt * 1e3, 3),
                              "clip_pairs_per_s": round(pairs / dt, 1), "arena_MB": round(loader.host[0].nbytes / 1e6, 1)}))


if __name__ == "__main__":
    main()
