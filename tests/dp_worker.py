"""One RANK of tests/test_gpu_dp_procs.py (not a test module): a real process of the data-parallel native step.

    python tests/dp_worker.py <rank> <world> <port> <out.npz> <seed> <cc_weight> <train> <device>

All ranks share ONE GPU (the test boxes have one), so the process group is "gloo" and dist.py stages its collectives through
host memory; the kernels, the C phase calls (coot_step_forward / coot_contrastive_fwd_bwd_dp / coot_step_backward) and the host
logic of RetrievalTrainer._train_step_native_dp are exactly those of an RCCL run (nntrainer/trainer_base.py:126-129 semantics:
encoders per shard, loss on the full batch).  device = "cpu" builds batch and shards only (the CPU suite's check of the harness).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DIMS = (64, 48, 64, 4, 64, 128)


def problem(seed, world):
    """The union batch (ragged), the per-rank shard bounds and the injected cycle-consistency positions — identical on every
    rank and in the parent (seeded)."""
    from oracle import coot_oracle as O
    rs = np.random.RandomState(seed)
    Bt = 4 * world + 1
    counts = rs.randint(1, 6, size=Bt).tolist()
    b = O.make_batch(seed + 31, Bt, counts, 12, 10, 9, 6, DIMS[0], DIMS[1], ragged=True, corr=0.5)
    idx_c = np.array([rs.randint(0, c) for c in counts], dtype=np.int64)
    idx_s = np.array([rs.randint(0, c) for c in counts], dtype=np.int64)
    cuts = np.sort(rs.choice(np.arange(1, Bt), size=world - 1, replace=False))
    vid_counts = np.diff(np.concatenate([[0], cuts, [Bt]])).astype(int).tolist()
    bounds = np.concatenate([[0], np.cumsum(vid_counts)]).astype(int)
    clip_counts = [int(sum(counts[bounds[r]:bounds[r + 1]])) for r in range(world)]
    return b, counts, idx_c, idx_s, vid_counts, clip_counts, bounds


def shard_numpy(b, counts, v0, v1):
    c0, c1 = int(sum(counts[:v0])), int(sum(counts[:v1]))
    out = {}
    for k, v in b.items():
        out[k] = v[c0:c1] if k.startswith(("clip_feat", "sent_feat")) else v[v0:v1]
    return out


def main():
    rank, world, port, out, seed, cc_weight, train, device = sys.argv[1:9]
    rank, world, seed, cc_weight, train = int(rank), int(world), int(seed), float(cc_weight), int(train)
    b, counts, idx_c, idx_s, vid_counts, clip_counts, bounds = problem(seed, world)
    sh = shard_numpy(b, counts, bounds[rank], bounds[rank + 1])
    if device == "cpu":
        np.savez(out, n_vid=len(sh["clip_num"]), n_clip=sh["clip_feat"].shape[0], clip_sum=int(np.sum(sh["clip_num"])))
        return
    import torch
    import torch.distributed as dist
    import coot_videotext_amd as cva
    from coot_videotext_amd import dist as cdist
    from oracle import coot_oracle as O
    from tests import helpers as H
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfgs = H.full_cfgs(*DIMS)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=cc_weight)
    mgr.set_all_models_train() if train else mgr.set_all_models_eval()
    tr = cva.RetrievalTrainer(cfg, mgr)
    tr.dp = cdist.DataParallelContext()
    # NaN-poison what the step's zero launch skips: the data-parallel backward must WRITE every weight-matrix gradient
    cva.lib.check(cva.lib.load().coot_set_option(b"grad_poison", 1), "grad_poison")
    batch = cva.synthetic.batch_from_numpy(sh)
    cc_idx = torch.from_numpy(np.concatenate([idx_c[bounds[rank]:bounds[rank + 1]], idx_s[bounds[rank]:bounds[rank + 1]]])).cuda()
    # counts NOT passed: the step learns the shard sizes and the global Cmax through its own collectives
    losses = tr.train_step_native(batch, do_optimizer=False, cc_indices=cc_idx)
    torch.cuda.synchronize()
    res = {f"g{i}": n._grad_flat.detach().cpu().numpy() for i, n in enumerate(mgr.model_dict.values())}
    res["losses"] = np.array([float(v) for v in losses])
    res["cmax"] = np.array([batch.max_clip_num, batch.max_sent_num])
    # a second step WITH the optimizer: parameters must stay identical across ranks (same all-reduced gradients, same update)
    tr.train_step_native(batch, do_optimizer=True, cc_indices=cc_idx)
    torch.cuda.synchronize()
    res["p0_after"] = list(mgr.model_dict.values())[0]._flat.detach().cpu().numpy()
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
