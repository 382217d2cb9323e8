"""Deterministic mode (include/coot_hip.h: coot_det_configure; RetrievalTrainer.set_deterministic) — the reference tests run-to-run
determinism of training (tests_nntrainer/integration_deter.py:18-66: two runs from the same seed give the same model).

The library's only order-dependent arithmetic is a handful of fp32 atomic accumulations; with the shipped Adam eps = 1e-8 their last-bit
noise is amplified into different trajectories (an update is +-lr whatever the gradient's size).  In deterministic mode those sums go
through order-independent fixed-point accumulators, so:
  * two runs of the same six optimizer steps end with BIT-IDENTICAL parameters and losses — with the shipped eps, on the fused kernels;
  * the re-orderings of the timed mode (next batch's input LayerNorm inside the previous step, deferred text join) reproduce the plain
    run's trajectory exactly, not to a tolerance;
  * one step's gradients equal the default mode's to fp32 round-off (the mode changes the order of a few sums, nothing else).
"""
import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

DIMS = (256, 192, 384, 8, 384, 768)  # d_model 384: the fused token-tile chains and the single-launch global networks


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _run(torch, cva, cfgs, Ps, batches, det, lookahead=False, defer=False, dp=None, steps=6):
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg_x, mgr)
    tr.lookahead_min_stage_bytes = 0
    tr.set_deterministic(det)
    if dp is not None:
        tr.dp = dp
    losses = []
    try:
        for it in range(steps):
            b = batches[it % len(batches)]
            nxt = batches[(it + 1) % len(batches)] if lookahead and it + 1 < steps else None
            kw = {}
            if dp is not None:
                b.global_max_synced = True
                if nxt is not None:
                    nxt.global_max_synced = True
                kw = dict(vid_counts=[int(b.clip_num.shape[0])], clip_counts=[int(b.clip_feat_len.shape[0])])
            out = tr.train_step_native(b, seed=500 + it, next_batch=nxt, defer_join=defer, **kw)
            tr.join_streams()
            losses.append([float(v) for v in out])
        torch.cuda.synchronize()
        return losses, [n._flat.detach().clone() for n in mgr.model_dict.values()], [n._grad_flat.detach().clone() for n in mgr.model_dict.values()]
    finally:
        tr.set_deterministic(False)


def _inject_nonfinite_addend(torch, cva, arena):
    """One non-finite addend through an accumulating kernel of the library into the first word of a registered arena: the
    cycle-consistency loss word of coot_cyclecons_fwd_bwd (csrc/loss.hip: acc_add) with an infinite loss weight."""
    lib = cva.lib.load()
    clip, sent = torch.randn(1, 2, 8, device="cuda"), torch.randn(1, 2, 8, device="cuda")
    lens, idx = torch.tensor([2], device="cuda"), torch.tensor([0], device="cuda")
    p = cva.lib.ptr
    cva.lib.check(lib.coot_cyclecons_fwd_bwd(p(clip), p(sent), p(lens), p(lens), p(idx), p(idx), 1, 2, 2, 8, float("inf"), 1.0, arena.data_ptr(),
                                             None, None, None, None, cva.lib.stream_ptr()), "coot_cyclecons_fwd_bwd")


def _batches(cva, ragged):
    if ragged:
        return [cva.synthetic.make_batch(40 + i, 12, cva.synthetic.anet_like_counts(70 + i, 12), 40, 40, 32, 16, DIMS[0], DIMS[1], ragged=True, packed=True)
                for i in range(3)]
    return [cva.synthetic.make_batch(40 + i, 12, 4, 40, 40, 32, 16, DIMS[0], DIMS[1], ragged=False) for i in range(3)]


@pytest.mark.parametrize("ragged", [False, True])
def test_two_runs_are_bit_identical_and_reorderings_change_nothing(env, ragged):
    torch, cva = env
    cfgs = H.full_cfgs(*DIMS)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batches = _batches(cva, ragged)
    la, pa, _ = _run(torch, cva, cfgs, Ps, batches, det=True)
    lb, pb, _ = _run(torch, cva, cfgs, Ps, batches, det=True)
    assert la == lb, (la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    # the timed mode's re-orderings: same trajectory, exactly
    lc, pc, _ = _run(torch, cva, cfgs, Ps, batches, det=True, lookahead=True, defer=True)
    assert la == lc, (la, lc)
    for a, c in zip(pa, pc):
        assert torch.equal(a, c)
    assert np.isfinite(np.array(la)).all() and la[0][0] != la[-1][0]


def test_deterministic_mode_computes_the_same_step(env):
    """One step (no update): gradients with and without the mode agree to fp32 round-off of the accumulated vectors."""
    torch, cva = env
    cfgs = H.full_cfgs(*DIMS)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batches = _batches(cva, False)
    res = []
    for det in (False, True):
        cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
        mgr.set_all_models_train()
        tr = cva.RetrievalTrainer(cfg_x, mgr)
        tr.set_deterministic(det)
        try:
            out = tr.train_step_native(batches[0], do_optimizer=False, seed=77)
            torch.cuda.synchronize()
            res.append(([float(v) for v in out], [n._grad_flat.detach().clone() for n in mgr.model_dict.values()]))
            # an addend outside the fixed-point range silently leaves the deterministic path — it is counted now
            assert tr.det_bypass_count() == (0 if det else -1)
            if det:  # one Inf gradient addend through an accumulating kernel of the library: counted, and visible in the fp32 word
                g = mgr.model_dict["net_video_global"]._grad_flat
                _inject_nonfinite_addend(torch, cva, g)
                torch.cuda.synchronize()
                assert tr.det_bypass_count() >= 1 and not bool(torch.isfinite(g[0]))
        finally:
            tr.set_deterministic(False)
            assert cva.loss_fn._DET_LOSS_WORD is None  # (cleared with the mode, not at the next autograd-route step)
    (l0, g0), (l1, g1) = res
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-9), (l0, l1)
    for a, b in zip(g0, g1):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale, (float((a - b).abs().max()), scale)
        assert float(b.abs().max()) > 0


def test_data_parallel_phase_path_is_deterministic_too(env):
    """The phase calls (coot_step_forward / loss on the exchanged block / coot_step_backward / bucketed reduction / coot_step_update)
    with a one-rank context: two runs bit-identical."""
    from tests.test_gpu_train_parity import _OneRankDP
    torch, cva = env
    cfgs = H.full_cfgs(*DIMS)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batches = _batches(cva, False)
    la, pa, _ = _run(torch, cva, cfgs, Ps, batches, det=True, dp=_OneRankDP(), steps=4)
    lb, pb, _ = _run(torch, cva, cfgs, Ps, batches, det=True, dp=_OneRankDP(), steps=4, lookahead=True, defer=True)
    assert la == lb, (la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def _plan_run(torch, cva, cfgs, Ps, batches, plan, use_graph):
    """One trainer, one native step per entry of `plan` (the deterministic flag of that step), seeds from the step counter (what a
    captured step can replay).  Returns (losses, parameters, number of cached graphs after every step)."""
    torch.manual_seed(4321)
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg_x, mgr)
    losses, cached = [], []
    try:
        for it, det in enumerate(plan):
            if det != bool(getattr(tr, "deterministic", False)):
                tr.set_deterministic(det)
                assert not getattr(getattr(tr, "_native", None), "graphs", None), "a change of the deterministic mode must drop every captured step"
            out = tr.train_step_native(batches[it % len(batches)], use_graph=use_graph)
            losses.append([float(v) for v in out])
            cached.append(len(getattr(tr._native, "graphs", {}) or {}))
        torch.cuda.synchronize()
        return losses, [n._flat.detach().clone() for n in mgr.model_dict.values()], cached
    finally:
        tr.set_deterministic(False)


def test_captured_steps_follow_the_deterministic_mode(env):
    """the library consults its process-wide deterministic table at RUN time, a captured step holds the flush launches
    (or their absence) and the shadow's address of the mode it was captured in.  RetrievalTrainer therefore keys its captured steps
    by the mode + configuration epoch, configures the mode before a capture and drops every cached graph when it changes:
      * deterministic + graph replay: two runs are bit-identical, and identical to the eager deterministic run (same launches);
      * toggling the mode between replays re-captures (a graph captured with the mode off has no flush nodes: replayed with the mode
        on it would drop the bias / LayerNorm gradients; one captured with it on reads a shadow that is freed when it goes off) and
        follows the eager run of the same plan to fp32 round-off."""
    torch, cva = env
    cfgs = H.full_cfgs(*DIMS)
    Ps = [O.make_params(cfgs[i], 3 + 10 * i) for i in range(4)]
    batches = _batches(cva, ragged=False)[:1]  # one shape: one captured step per mode
    det_plan = [True] * 5
    l1, p1, c1 = _plan_run(torch, cva, cfgs, Ps, batches, det_plan, use_graph=True)
    l2, p2, _ = _plan_run(torch, cva, cfgs, Ps, batches, det_plan, use_graph=True)
    le, pe, _ = _plan_run(torch, cva, cfgs, Ps, batches, det_plan, use_graph=False)
    assert c1[-1] == 1, c1  # (step 0 runs eagerly and creates the lazy state, step 1 captures, the rest replay)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(p1, p2)), "deterministic graph replays differ between two runs"
    assert l1 == le and all(torch.equal(a, b) for a, b in zip(p1, pe)), "deterministic graph replay differs from the eager deterministic run"
    plan = [False, False, False, True, True, True, False, False]
    lg, pg, cg = _plan_run(torch, cva, cfgs, Ps, batches, plan, use_graph=True)
    lq, pq, _ = _plan_run(torch, cva, cfgs, Ps, batches, plan, use_graph=False)
    assert cg == [0, 1, 1, 1, 1, 1, 1, 1], cg  # (dropped at each change of the mode — asserted in _plan_run — and captured again by the next step)
    for a, b in zip(lg, lq):  # (one batch trained for 8 steps: by then the loss is a few hinge violations, sensitive to the last bits the
        assert np.allclose(a, b, rtol=3e-2, atol=3e-3), (a, b)  # non-deterministic steps differ in; a dropped gradient or a freed shadow moves it by O(1))
    for a, b in zip(pg, pq):
        assert torch.isfinite(a).all()
        # (Adam at eps = 1e-8 turns last-bit differences of the non-deterministic steps into +-lr per entry: compare at that scale)
        assert float((a - b).abs().max()) <= 8 * 2e-3, float((a - b).abs().max())
