"""Multi-process data-parallel logic on CPU (gloo, world_size 2): the collectives of
coot_videotext_amd/dist.py and the equivalence "shard -> gather embeddings -> full-batch loss on every rank -> own-row
gradients -> all-reduce(SUM) of parameter gradients" == single-process full-batch gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _contrastive(im, s, margin=0.2):
    """ContrastiveLoss.forward in plain torch (coot/loss_fn.py:63-100) — stand-in for the HIP loss on CPU."""
    im, s = torch.nn.functional.normalize(im), torch.nn.functional.normalize(s)
    scores = im @ s.t()
    d = scores.diag().view(-1, 1)
    eye = torch.eye(scores.shape[0], dtype=torch.bool)
    cs = (margin + scores - d).clamp(min=0).masked_fill(eye, 0)
    ci = (margin + scores - d.t()).clamp(min=0).masked_fill(eye, 0)
    return (cs.sum() + ci.sum()) / (scores.shape[0] ** 2)


def _worker(rank, world, port, counts_v, counts_c, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import coot_videotext_amd  # noqa: F401  (registers the package alias)
    from coot_videotext_amd import dist as cdist
    dp = cdist.DataParallelContext()
    torch.manual_seed(0)
    enc_v, enc_t = torch.nn.Linear(6, 8), torch.nn.Linear(5, 8)  # identical init on every rank
    g = torch.Generator().manual_seed(1)
    Nv, Nc = sum(counts_v), sum(counts_c)
    xv, xt = torch.randn(Nv, 6, generator=g), torch.randn(Nv, 5, generator=g)
    xc, xs = torch.randn(Nc, 6, generator=g), torch.randn(Nc, 5, generator=g)
    v0, c0 = sum(counts_v[:rank]), sum(counts_c[:rank])
    sl_v, sl_c = slice(v0, v0 + counts_v[rank]), slice(c0, c0 + counts_c[rank])

    class E:  # the six fields gather_embeddings reads
        pass

    vis, txt = E(), E()
    vis.vid_emb, vis.vid_context, vis.clip_emb = enc_v(xv[sl_v]).repeat(1, 2), enc_v(xv[sl_v]), enc_v(xc[sl_c])
    txt.par_emb, txt.par_context, txt.sent_emb = enc_t(xt[sl_v]).repeat(1, 2), enc_t(xt[sl_v]), enc_t(xs[sl_c])
    # counts discovered through the collective path (no host-side knowledge)
    vid_emb, par_emb, clip_emb, sent_emb, vctx, pctx, gb = dp.gather_embeddings(vis, txt)
    assert gb == Nv and vid_emb.shape == (Nv, 16) and clip_emb.shape == (Nc, 8)
    loss = _contrastive(vid_emb, par_emb) + _contrastive(clip_emb, sent_emb) + _contrastive(vctx, pctx)
    loss.backward()
    grads = [p.grad.clone() for p in list(enc_v.parameters()) + list(enc_t.parameters())]
    flat = torch.cat([gr.reshape(-1) for gr in grads])
    dp.allreduce_grads([flat])
    # helpers
    assert dp.global_max(3 + rank, "cpu") == 3 + world - 1
    assert dp.global_counts(counts_c[rank], "cpu") == list(counts_c)
    assert dp.global_max_pair(3 + rank, 10 - rank, "cpu") == (3 + world - 1, 10)
    # the batch-shape exchange of the native step: ONE host collective, every rank sees every rank's integers in rank order
    assert dp.exchange_shapes([5 + rank, 20 - rank, 7]) == [[5 + r, 20 - r, 7] for r in range(world)]
    blk = torch.full((6,), float(rank))
    got = torch.empty(world * 6)
    dp.gather_block(blk, got)   # the one-block embedding exchange
    assert got.view(world, 6).eq(torch.arange(world, dtype=torch.float32)[:, None]).all()
    # RetrievalTrainer._dp_batch_shapes: what a rank learns about the global batch before it sizes its step — every rank's videos / clips
    # and the GLOBAL max clips / sentences per video (padding to it keeps avg_special pooling reference-exact) — from host integers only
    import types
    from coot_videotext_amd.trainer_retrieval import RetrievalTrainer
    stub = types.SimpleNamespace(dp=dp)
    batch = types.SimpleNamespace(clip_num=torch.zeros(counts_v[rank], dtype=torch.int64), clip_feat_len=torch.zeros(counts_c[rank], dtype=torch.int64),
                                  max_clip_num=3 + rank, max_sent_num=9 - 2 * rank)
    vc, cc_ = RetrievalTrainer._dp_batch_shapes(stub, batch)
    assert vc == list(counts_v) and cc_ == list(counts_c)
    assert (batch.max_clip_num, batch.max_sent_num) == (3 + world - 1, 9) and batch.global_max_synced
    assert RetrievalTrainer._dp_batch_shapes(stub, batch, [1, 2], [3, 4]) == ([1, 2], [3, 4])   # fixed shapes given + synced: no collective
    # the collective of the native data-parallel step (no autograd): ragged row blocks arrive in rank order
    rows = torch.arange(counts_c[rank] * 2, dtype=torch.float32).view(-1, 2) + 100 * rank
    allrows = cdist.gather_rows_nograd(rows, list(counts_c))
    exp = torch.cat([torch.arange(c * 2, dtype=torch.float32).view(-1, 2) + 100 * r for r, c in enumerate(counts_c)])
    assert torch.equal(allrows, exp)
    if rank == 0:
        # single-process reference on the full batch
        torch.manual_seed(0)
        rv, rt = torch.nn.Linear(6, 8), torch.nn.Linear(5, 8)
        l2 = (_contrastive(rv(xv).repeat(1, 2), rt(xt).repeat(1, 2)) + _contrastive(rv(xc), rt(xs)) + _contrastive(rv(xv), rt(xt)))
        l2.backward()
        ref = torch.cat([p.grad.reshape(-1) for p in list(rv.parameters()) + list(rt.parameters())])
        out.put((float(loss), float(l2), float((flat - ref).abs().max()), float(ref.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts_v,counts_c", [((3, 3), (7, 7)), ((4, 2), (9, 5))])
def test_data_parallel_equals_full_batch(counts_v, counts_c):
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts_v, counts_c, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    loss, ref_loss, err, scale = out.get()
    assert abs(loss - ref_loss) < 1e-6
    assert err < 1e-5 * max(scale, 1.0), (err, scale)


def test_gather_rows_backward_is_own_slice():
    """Single-process sanity (world 1, gloo) of the autograd contract of gather_rows."""
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import coot_videotext_amd  # noqa: F401
        from coot_videotext_amd import dist as cdist
        x = torch.randn(5, 3, requires_grad=True)
        y = cdist.gather_rows(x, [5], 0)
        (y * torch.arange(15.0).view(5, 3)).sum().backward()
        assert torch.equal(x.grad, torch.arange(15.0).view(5, 3))
    finally:
        dist.destroy_process_group()


def _rccl_worker(rank, world, port, break_rank, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import coot_videotext_amd  # noqa: F401
    from coot_videotext_amd import dist as cdist
    # break_rank cannot even load the library: the others must not enter the (collective) communicator init alone
    r = cdist.DirectRccl(lib_path="/nonexistent/librccl.so" if rank == break_rank else None)
    out.put((rank, r.comm is None, r.lib is not None))
    # the torch group is still usable afterwards (the fallback route): one collective to prove it
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    assert float(t) == 3.0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("break_rank", [-1, 1])
def test_direct_rccl_creation_is_agreed_between_ranks(break_rank):
    """dist.DirectRccl (the step's collectives as direct RCCL calls) is created COLLECTIVELY: library load and communicator init are
    agreed over the torch group, so a rank that fails never leaves the others inside ncclCommInitRank and every rank ends on the same
    route.  Two gloo ranks without a GPU: the unique id is really created and broadcast, the init fails on both (no device) — or one
    rank cannot load the library at all — and both come back with no communicator instead of hanging."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, break_rank, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = sorted(out.get() for _ in range(2))
    assert all(no_comm for _, no_comm, _ in res)
    assert [lib for _, _, lib in res] == [True, break_rank != 1]
