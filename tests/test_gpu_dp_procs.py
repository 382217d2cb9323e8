"""The data-parallel native step with REAL processes (replaces nn.DataParallel of nntrainer/trainer_base.py:126-129).

Two / three ranks, each its own process with its own HIP context, library state and RetrievalTrainer, exchange embeddings and
gradients through torch.distributed ("gloo": all ranks share the one GPU of the test box, RCCL refuses that; dist.py stages the
collectives through host memory — everything else is the production path).  Ragged shards (rank sizes differ, clip counts
differ), shard sizes and the global Cmax learnt through the step's own collectives.  Checked on EVERY rank, rank 1 included:
the all-reduced parameter gradients and the reduced losses equal the single-GPU native step on the union batch; after an
optimizer step the parameters are identical on all ranks.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import dp_worker as W
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(world, tmp_path, seed, cc_weight, device):
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), str(port), outs[r], str(seed),
                               str(cc_weight), "1", device], cwd=ROOT) for r in range(world)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world, rcs
    return [dict(np.load(o)) for o in outs]


def test_worker_shards_cover_the_union_batch(tmp_path):
    """CPU: the harness itself — the workers' shards partition the union batch (video-major collation)."""
    world, seed = 3, 5
    res = _run_ranks(world, tmp_path, seed, 0.01, "cpu")
    b, counts, _, _, vid_counts, clip_counts, _ = W.problem(seed, world)
    assert [int(r["n_vid"]) for r in res] == vid_counts and sum(vid_counts) == len(counts)
    assert [int(r["n_clip"]) for r in res] == clip_counts == [int(r["clip_sum"]) for r in res]
    assert len(set(vid_counts)) > 1 or len(set(clip_counts)) > 1  # ragged shards


@pytest.mark.gpu
@pytest.mark.parametrize("world,cc_weight", [(2, 0.01), (3, 0.0)])
def test_native_dp_step_real_processes_match_union_batch(tmp_path, world, cc_weight):
    import torch
    import coot_videotext_amd as cva
    from oracle import coot_oracle as O
    seed = 11 * world
    res = _run_ranks(world, tmp_path, seed, cc_weight, "cuda")
    # the single-GPU native step on the union batch, in this process
    b, counts, idx_c, idx_s, vid_counts, clip_counts, _ = W.problem(seed, world)
    cfgs = H.full_cfgs(*W.DIMS)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=cc_weight)
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg, mgr)
    union = cva.synthetic.batch_from_numpy(b)
    la = [float(v) for v in tr.train_step_native(union, do_optimizer=False, cc_indices=torch.from_numpy(np.concatenate([idx_c, idx_s])).cuda())]
    torch.cuda.synchronize()
    g_ref = [n._grad_flat.detach().cpu().numpy() for n in mgr.model_dict.values()]
    for r, out in enumerate(res):
        assert tuple(out["cmax"]) == (max(counts), max(counts)), (r, out["cmax"])  # padded to the GLOBAL Cmax (avg_special parity)
        l = out["losses"]
        assert abs(l[1] - la[1]) < 1e-5 * max(1.0, abs(la[1])) and abs(l[2] - la[2]) < 1e-5 * max(1.0, abs(la[2])) + 1e-8, (r, l, la)
        assert abs(l[0] - la[0]) < 2e-5 * max(1.0, abs(la[0]))
        for i, name in enumerate(H.NET_KEYS):
            scale = float(np.abs(g_ref[i]).max())
            err = float(np.abs(out[f"g{i}"] - g_ref[i]).max()) / scale
            assert err < 2e-4, (r, name, err)  # fp32 summation order only
    for out in res[1:]:
        assert np.array_equal(out["p0_after"], res[0]["p0_after"])  # same update everywhere: replicas stay bit-identical
