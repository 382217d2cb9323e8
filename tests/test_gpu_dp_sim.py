"""Data-parallel path with SIMULATED ranks on one GPU (what nn.DataParallel does implicitly in
nntrainer/trainer_base.py:126-129: encoders per shard, loss on the full batch).

(1) coot_contrastive_fwd_bwd_dp, called once per simulated rank with its window (own_high0, own_high, own_low0, own_low) on
    strided views of the gathered buffers — the ranks' loss shares add up to the single-GPU loss and the concatenation of the
    own-row gradients equals the full-batch gradient, for R in {2, 4, 8} and ragged shard sizes.
(2) RetrievalTrainer._train_step_native_dp driven rank by rank through a fake DataParallelContext whose collectives are
    served from a blackboard (all-gather = concatenation of the shards' rows, all-reduce = the test sums the partial
    results) — the summed parameter gradients and losses equal the single-GPU native step on the union batch (dropout 0,
    cycle-consistency positions injected so both draw the same ones).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _split(total, R, rs):
    """R positive shard sizes summing to total, ragged."""
    cuts = np.sort(rs.choice(np.arange(1, total), size=R - 1, replace=False))
    return np.diff(np.concatenate([[0], cuts, [total]])).astype(int).tolist()


@pytest.mark.parametrize("R,nh0", [(2, 24), (4, 24), (8, 24), (4, 330), (8, 600)])
def test_contrastive_dp_windows_reproduce_full_batch(env, R, nh0):
    """(nh0 = 330 / 600 videos: ~1 300 / ~2 400 clips — strips as long as those of 8 ranks of the ActivityNet workload.)"""
    torch, cva = env
    lib, L = cva.lib.load(), cva.lib
    rs = np.random.RandomState(100 + R + nh0)
    D = 64
    nh = nh0 + R
    vid_counts = _split(nh, R, rs)
    clips_per_video = rs.randint(1, 6, size=nh)
    clip_counts = [int(clips_per_video[sum(vid_counts[:r]):sum(vid_counts[:r + 1])].sum()) for r in range(R)]
    nl = int(clips_per_video.sum())
    # correlated sets so that margins are violated (independent random vectors never violate them)
    base_h, base_l = rs.randn(1, 2 * D), rs.randn(1, D)
    E = {}
    E["vid"] = base_h + 0.6 * rs.randn(nh, 2 * D); E["par"] = E["vid"] + 0.3 * rs.randn(nh, 2 * D)
    E["clip"] = base_l + 0.6 * rs.randn(nl, D); E["sent"] = E["clip"] + 0.3 * rs.randn(nl, D)
    E["vctx"] = base_l + 0.6 * rs.randn(nh, D); E["pctx"] = E["vctx"] + 0.3 * rs.randn(nh, D)
    w = dict(H.ANET_W, weight_context_internal=0.5)
    cfg = cva.ContrastiveLossConfig(0.2, **w).to_c()
    dev = "cuda"
    t = {k: torch.from_numpy(v).float().to(dev).contiguous() for k, v in E.items()}
    order = ("vid", "par", "clip", "sent", "vctx", "pctx")
    sp = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(lib.coot_contrastive_scratch_bytes(nh, nl, 2 * D, D), dtype=torch.uint8, device=dev)
    # single GPU
    loss1 = torch.zeros(1, device=dev)
    g1 = {k: torch.zeros_like(t[k]) for k in order}
    L.check(lib.coot_contrastive_fwd_bwd(C.byref(cfg), nh, nl, 2 * D, D, *[t[k].data_ptr() for k in order], loss1.data_ptr(),
                                         *[g1[k].data_ptr() for k in order], scratch.data_ptr(), scratch.numel(), sp), "contrastive")
    # oracle, for the absolute scale
    loss_o, dE = O.total_contrastive_loss(dict(vid_emb=E["vid"], par_emb=E["par"], clip_emb=E["clip"], sent_emb=E["sent"],
                                               vid_context=E["vctx"], par_context=E["pctx"]), w, 0.2)
    torch.cuda.synchronize()
    assert abs(float(loss1) - loss_o) < 5e-3
    if nh0 >= 330:  # (the strips of these batches are split over several workgroups; at 26 rows one hinge term at the margin, decided on
        for k, ko in zip(order, ("vid_emb", "par_emb", "clip_emb", "sent_emb", "vid_context", "par_context")):  # bf16 scores, moves the cosine by 5e-3)
            assert H.cosine_flat(g1[k].cpu().numpy(), dE[ko]) > 0.995, k
    # the gathered buffers as the trainer lays them out: [n, 2D | 2D | D | D] and [n, D | D]
    high = torch.cat([t["vid"], t["par"], t["vctx"], t["pctx"]], dim=1).contiguous()
    low = torch.cat([t["clip"], t["sent"]], dim=1).contiguous()
    hp, lp, e4 = high.data_ptr(), low.data_ptr(), 4
    sets = (C.c_void_p * 6)(hp, hp + 2 * D * e4, lp, lp + D * e4, hp + 4 * D * e4, hp + 5 * D * e4)
    lds = (C.c_int64 * 6)(6 * D, 6 * D, 2 * D, 2 * D, 6 * D, 6 * D)
    got = {k: [] for k in order}
    loss_sum = 0.0
    for r in range(R):
        v0, c0, B, Nc = sum(vid_counts[:r]), sum(clip_counts[:r]), vid_counts[r], clip_counts[r]
        own = {"vid": torch.zeros(B, 2 * D, device=dev), "par": torch.zeros(B, 2 * D, device=dev), "clip": torch.zeros(Nc, D, device=dev),
               "sent": torch.zeros(Nc, D, device=dev), "vctx": torch.zeros(B, D, device=dev), "pctx": torch.zeros(B, D, device=dev)}
        down = (C.c_void_p * 6)(*[own[k].data_ptr() for k in order])
        loss_r = torch.zeros(1, device=dev)
        L.check(lib.coot_contrastive_fwd_bwd_dp(C.byref(cfg), nh, nl, 2 * D, D, C.byref(sets), C.byref(lds), loss_r.data_ptr(), C.byref(down),
                                                v0, B, c0, Nc, scratch.data_ptr(), scratch.numel(), sp), "contrastive_dp")
        torch.cuda.synchronize()
        loss_sum += float(loss_r)   # a rank's call returns ITS share of the loss (the hinge terms of its rows)
        assert float(loss_r) >= 0.0
        for k in order:
            got[k].append(own[k])
    assert abs(loss_sum - float(loss1)) <= 2e-6 * max(1.0, abs(float(loss1))), (loss_sum, float(loss1))
    for k in order:
        full = torch.cat(got[k], dim=0)
        assert full.shape == g1[k].shape
        err = float((full - g1[k]).abs().max()) / max(float(g1[k].abs().max()), 1e-30)
        assert err < 1e-5, (k, err)


def test_contrastive_column_splits_agree(env):
    """cl_half splits the columns of a 16-row strip over several workgroups when the (gathered) batch is large; the partial
    gradient strips, violation counts and hinge sums are added by cl_finish.  Forced splits at a small batch: same loss and
    gradients as the unsplit strip (the hinge matrix is identical, only the fp32 summation order differs), and the oracle's."""
    torch, cva = env
    lib, L = cva.lib.load(), cva.lib
    rs = np.random.RandomState(7)
    D, nh, nl = 64, 100, 300
    base_h, base_l = rs.randn(1, 2 * D), rs.randn(1, D)
    E = {}
    E["vid"] = base_h + 0.6 * rs.randn(nh, 2 * D); E["par"] = E["vid"] + 0.3 * rs.randn(nh, 2 * D)
    E["clip"] = base_l + 0.6 * rs.randn(nl, D); E["sent"] = E["clip"] + 0.3 * rs.randn(nl, D)
    E["vctx"] = base_l + 0.6 * rs.randn(nh, D); E["pctx"] = E["vctx"] + 0.3 * rs.randn(nh, D)
    w = dict(H.ANET_W, weight_context_internal=0.5)
    cfg = cva.ContrastiveLossConfig(0.2, **w).to_c()
    order = ("vid", "par", "clip", "sent", "vctx", "pctx")
    t = {k: torch.from_numpy(v).float().cuda().contiguous() for k, v in E.items()}
    sp = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for cs in (1, 2, 3, 8):
            assert lib.coot_set_option(b"cl_col_split", cs) == 0
            scratch = torch.empty(lib.coot_contrastive_scratch_bytes(nh, nl, 2 * D, D), dtype=torch.uint8, device="cuda")
            loss = torch.zeros(1, device="cuda")
            g = {k: torch.zeros_like(t[k]) for k in order}
            L.check(lib.coot_contrastive_fwd_bwd(C.byref(cfg), nh, nl, 2 * D, D, *[t[k].data_ptr() for k in order], loss.data_ptr(),
                                                 *[g[k].data_ptr() for k in order], scratch.data_ptr(), scratch.numel(), sp), "contrastive")
            torch.cuda.synchronize()
            res[cs] = (float(loss), g)
    finally:
        lib.coot_set_option(b"cl_col_split", 0)
    loss_o, dE = O.total_contrastive_loss(dict(vid_emb=E["vid"], par_emb=E["par"], clip_emb=E["clip"], sent_emb=E["sent"],
                                               vid_context=E["vctx"], par_context=E["pctx"]), w, 0.2)
    assert abs(res[1][0] - loss_o) < 5e-3
    for cs in (2, 3, 8):
        assert abs(res[cs][0] - res[1][0]) <= 2e-6 * max(1.0, abs(res[1][0])), (cs, res[cs][0], res[1][0])
        for k, ko in zip(order, ("vid_emb", "par_emb", "clip_emb", "sent_emb", "vid_context", "par_context")):
            err = float((res[cs][1][k] - res[1][1][k]).abs().max()) / max(float(res[1][1][k].abs().max()), 1e-30)
            assert err < 1e-5, (cs, k, err)
            assert H.cosine_flat(res[cs][1][k].cpu().numpy(), dE[ko]) > 0.99, (cs, k)


def _slice_batch(cva, torch, b, v0, v1, counts):
    """Videos [v0, v1) of a RetrievalDataBatchTuple with their clips / sentences (collation is video-major)."""
    c0, c1 = int(sum(counts[:v0])), int(sum(counts[:v1]))
    f = {}
    for k in ("vid_feat", "vid_feat_mask", "vid_feat_len", "par_feat", "par_feat_mask", "par_feat_len", "clip_num", "sent_num"):
        f[k] = getattr(b, k)[v0:v1].contiguous()
    for k in ("clip_feat", "clip_feat_mask", "clip_feat_len", "sent_feat", "sent_feat_mask", "sent_feat_len"):
        f[k] = getattr(b, k)[c0:c1].contiguous()
    keys = [str(i) for i in range(v0, v1)]
    return cva.RetrievalDataBatchTuple(key=keys, data_key=keys, sentences=[[""]] * (v1 - v0), max_clip_num=None, max_sent_num=None, **f)


class _Blackboard:
    """What the simulated collectives exchange: rows every rank contributed to each all-gather of a step (by call order)."""

    def __init__(self, R):
        self.R = R
        self.rows = {}       # (call index, rank) -> tensor
        self.global_max = None


class _FakeDP:
    """dist.DataParallelContext for ONE simulated rank: all-gathers are served from the blackboard (the other ranks' rows were
    recorded in an earlier pass over the same deterministic forward), all-reduces leave the rank's partial result in place
    (the test sums them)."""

    def __init__(self, rank, board):
        self.rank, self.world, self.group, self.board = rank, board.R, None, board
        self.calls = 0

    def global_max_pair(self, a, b, device):
        return self.board.global_max

    def global_counts(self, n, device):
        raise AssertionError("the test passes the counts")

    def gather_rows_nograd(self, x, counts):
        import torch
        i = self.calls
        self.calls += 1
        self.board.rows[(i, self.rank)] = x.detach().clone()
        parts = []
        for r in range(self.world):
            t = self.board.rows.get((i, r))
            parts.append(t if t is not None else x.new_zeros((counts[r],) + tuple(x.shape[1:])))
            assert parts[-1].shape[0] == counts[r]
        return torch.cat(parts, dim=0)

    def all_reduce_sum(self, t):
        pass


class _FakeDPBlocks(_FakeDP):
    """+ the one-block exchange (dist.DataParallelContext.gather_block): the step then reads the gathered blocks in place through
    coot_contrastive_fwd_bwd_dp_blocks instead of two packed row gathers."""

    def gather_block(self, send, recv):
        i = self.calls
        self.calls += 1
        self.board.rows[(i, self.rank)] = send.detach().clone()
        n = send.numel()
        for r in range(self.world):
            t = self.board.rows.get((i, r))
            if t is None:
                recv[r * n:(r + 1) * n].zero_()
            else:
                recv[r * n:(r + 1) * n].copy_(t)


@pytest.mark.parametrize("R,cc_weight,blocks", [(2, 0.01, True), (4, 0.0, True), (3, 0.01, True), (3, 0.01, False)])
def test_native_dp_step_simulated_ranks_match_union_batch(env, R, cc_weight, blocks):
    torch, cva = env
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    rs = np.random.RandomState(7 * R)
    Bt = 4 * R + 1
    counts = rs.randint(1, 6, size=Bt).tolist()
    union = cva.synthetic.batch_from_numpy(O.make_batch(31, Bt, counts, 12, 10, 9, 6, dims[0], dims[1], ragged=True, corr=0.5))
    idx_c = torch.tensor([rs.randint(0, c) for c in counts], dtype=torch.int64)
    idx_s = torch.tensor([rs.randint(0, c) for c in counts], dtype=torch.int64)
    # single-GPU native step on the union batch
    cfg_a, mgr_a = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=cc_weight)
    mgr_a.set_all_models_train()
    ta = cva.RetrievalTrainer(cfg_a, mgr_a)
    la = [float(v) for v in ta.train_step_native(union, do_optimizer=False, cc_indices=torch.cat([idx_c, idx_s]).cuda())]
    torch.cuda.synchronize()
    g_ref = [n._grad_flat.detach().clone() for n in mgr_a.model_dict.values()]
    # simulated ranks
    vid_counts = _split(Bt, R, rs)
    bounds = np.concatenate([[0], np.cumsum(vid_counts)]).astype(int)
    clip_counts = [int(sum(counts[bounds[r]:bounds[r + 1]])) for r in range(R)]
    board = _Blackboard(R)
    board.global_max = (max(counts), max(counts))
    cfg_b, mgr_b = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=cc_weight)
    mgr_b.set_all_models_train()
    tb = cva.RetrievalTrainer(cfg_b, mgr_b)
    shards = [_slice_batch(cva, torch, union, bounds[r], bounds[r + 1], counts) for r in range(R)]
    g_sum, loss_contr, loss_cc = None, [], 0.0
    for pass_ in range(2):  # pass 0 fills the blackboard (every rank's forward is deterministic), pass 1 is the step proper
        for r in range(R):
            tb.dp = (_FakeDPBlocks if blocks else _FakeDP)(r, board)
            sh = shards[r]
            sh.global_max_synced = False
            cc_idx = torch.cat([idx_c[bounds[r]:bounds[r + 1]], idx_s[bounds[r]:bounds[r + 1]]]).cuda()
            out = tb.train_step_native(sh, do_optimizer=False, vid_counts=vid_counts, clip_counts=clip_counts, cc_indices=cc_idx)
            torch.cuda.synchronize()
            assert (sh.max_clip_num, sh.max_sent_num) == board.global_max  # padded to the GLOBAL Cmax (avg_special parity)
            if pass_ == 1:
                st = tb._native
                assert st.blocks_on == blocks
                g = [n._grad_flat.detach().clone() for n in mgr_b.model_dict.values()]
                g_sum = g if g_sum is None else [a + b for a, b in zip(g_sum, g)]
                loss_contr.append(float(out[1]))
                loss_cc += float(st.cc_word)  # this rank's part of the global mean (the all-reduce would have summed them)
    # every rank computed ITS share of the full-batch contrastive loss (the all-reduce would have summed them, like the cc word)
    assert abs(sum(loss_contr) - la[1]) < 1e-5 * max(1.0, abs(la[1])), (loss_contr, la)
    assert abs(loss_cc - la[2]) < 1e-5 * max(1.0, abs(la[2])) + 1e-8, (loss_cc, la[2])
    for name, a, b in zip(H.NET_KEYS, g_sum, g_ref):
        scale = float(b.abs().max())
        err = float((a - b).abs().max()) / scale
        assert err < 2e-4, (name, err)  # fp32 atomics / split order only
