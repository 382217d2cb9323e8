"""Concurrent streams (include/coot_hip.h: coot_stream_create_concurrent / coot_streams_overlap).  HIP maps a process's streams onto
4 hardware queues in creation order; two streams on one queue serialise.  The step's side streams are therefore VERIFIED to overlap:
with one more stream created anywhere in the process before the trainer's, streams taken on trust put both sides of the step in one
queue (1.72 instead of 1.22 ms, profiles/r06_stream_queues.txt)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    return torch, cva


def _opt(cva, name):
    v = C.c_int(0)
    cva.lib.check(cva.lib.load().coot_get_option(name.encode(), C.byref(v)), name)
    return v.value


def test_overlap_test_sees_shared_queues(env):
    """The hazard is real and the test sees it: among nine streams taken from torch's pool at least two share one of the 4 hardware
    queues (their spin kernels run one after the other), and a stream never overlaps itself."""
    torch, cva = env
    lib = cva.lib.load()
    streams = [torch.cuda.Stream() for _ in range(9)]
    for s in streams:
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    assert lib.coot_streams_overlap(streams[0].cuda_stream, streams[0].cuda_stream) == 0
    res = {(i, j): lib.coot_streams_overlap(streams[i].cuda_stream, streams[j].cuda_stream) for i in range(9) for j in range(i + 1, 9)}
    assert all(r in (0, 1) for r in res.values()), res
    assert any(r == 1 for r in res.values()), "no two streams run concurrently"
    if not any(r == 0 for r in res.values()):  # (GPU_MAX_HW_QUEUES raised above the default 4: nothing to see on this box)
        pytest.skip("nine streams and every pair concurrent: more hardware queues than the default here")


@pytest.mark.parametrize("before", [0, 1, 2, 3])
def test_created_streams_overlap_whatever_was_created_before(env, before):
    torch, cva = env
    lib = cva.lib.load()
    junk = [torch.cuda.Stream() for _ in range(before)]
    for s in junk:
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    main = torch.cuda.current_stream()
    unresolved0 = _opt(cva, "stream_unresolved")  # (per-thread totals: other modules ran before this one)
    a = cva.lib.ConcurrentStream([main])
    b = cva.lib.ConcurrentStream([main, a])
    c = cva.lib.ConcurrentStream([main, a, b])
    try:
        assert a.concurrent and b.concurrent and c.concurrent
        hs = [main.cuda_stream, a.cuda_stream, b.cuda_stream, c.cuda_stream]
        for i in range(4):
            for j in range(i + 1, 4):
                assert lib.coot_streams_overlap(hs[i], hs[j]) == 1, (i, j)
        with torch.cuda.stream(a.torch):  # usable from torch
            x = torch.ones(8, device="cuda") * 2
        torch.cuda.synchronize()
        assert float(x.sum()) == 16.0
    finally:
        for s in (a, b, c):
            s.close()
    assert _opt(cva, "stream_unresolved") == unresolved0
    assert lib.coot_stream_destroy(C.c_void_p(12345)) != 0  # not one of ours


@pytest.mark.parametrize("before", [0, 1, 2, 3])
def test_step_streams_are_concurrent_in_every_creation_order(env, before):
    """The trainer's text stream overlaps the caller's stream and the library's own stream finds a queue beside both, whatever number
    of unrelated streams the process created first; the step's result does not depend on it."""
    torch, cva = env
    from oracle import coot_oracle as O
    from tests import helpers as H
    lib = cva.lib.load()
    junk = [torch.cuda.Stream() for _ in range(before)]
    for s in junk:
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batch = cva.synthetic.make_batch(7, 6, [1, 2, 3, 4, 2, 1], 12, 10, 9, 6, dims[0], dims[1], ragged=True)
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.0)
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg, mgr)
    unresolved0 = _opt(cva, "stream_unresolved")
    losses = tr.train_step_native(batch, do_optimizer=False)
    torch.cuda.synchronize()
    st = tr._native
    assert st.text_cs.concurrent
    assert lib.coot_streams_overlap(torch.cuda.current_stream().cuda_stream, st.streams[1].cuda_stream) == 1
    assert _opt(cva, "stream_unresolved") == unresolved0
    val = float(losses[0])
    assert val == val and val > 0
    # a caller that comes with another current stream gets a text stream verified against THAT one
    other = torch.cuda.Stream()
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):
        l2 = tr.train_step_native(batch, do_optimizer=False)
        torch.cuda.synchronize()
        assert lib.coot_streams_overlap(other.cuda_stream, tr._native.streams[1].cuda_stream) == 1
    assert abs(float(l2[0]) - val) <= 1e-6 * max(1.0, abs(val))
    tr.close()


@pytest.mark.parametrize("counts", [[1, 2, 3, 4, 2, 1], [4, 4, 4, 4, 4, 4], [1, 1, 5, 1, 2, 3]])
def test_pool_kernel_packs_items_like_the_pack_launch(env, counts):
    """The local network's pooling kernel writes the item embeddings into the global network's padded [B, Cmax, D] layout itself
    (csrc/pool.h: pk_*; coot_set_option("pool_handover", 0) = the stand-alone pack launch of coot/model_retrieval.py:121-136's loop).
    Both must give the same bits on a ragged batch — packed rows, ZERO padding rows (the buffers are NaN-poisoned first) and, through
    the mask and the lengths, the global network's outputs."""
    torch, cva = env
    from oracle import coot_oracle as O
    from tests import helpers as H
    lib = cva.lib.load()
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batch = cva.synthetic.make_batch(7, len(counts), counts, 12, 10, 9, 6, dims[0], dims[1], ragged=True)
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.0)
    mgr.set_all_models_eval()
    tr = cva.RetrievalTrainer(cfg, mgr)
    tr.train_step_native(batch, do_optimizer=False)  # (sets the native state up)
    st = tr._native
    _, x = tr._native_setup(batch)
    B, Nc, D = st.dims.B, st.dims.Nc, cfgs[0]["hidden_dim"] if isinstance(cfgs[0], dict) else st.cfg.net[0].hidden_dim
    dev = batch.vid_feat.device
    outs = {}
    try:
        for mode in (0, 1):
            cva.lib.check(lib.coot_set_option(b"pool_handover", mode), "pool_handover")
            t = [torch.full((B + Nc, D), float("nan"), device=dev), torch.full((B + Nc, D), float("nan"), device=dev),
                 torch.full((B, 2 * D), float("nan"), device=dev), torch.full((B, 2 * D), float("nan"), device=dev),
                 torch.full((B, st.dims.Cmax_clip, D), float("nan"), device=dev), torch.full((B, st.dims.Cmax_sent, D), float("nan"), device=dev)]
            main = torch.cuda.current_stream()
            cva.lib.check(lib.coot_step_forward(C.byref(st.cfg), C.byref(st.bufs), C.byref(x), C.byref(st.dims), *[v.data_ptr() for v in t],
                                                st.ws.data_ptr(), st.ws.numel(), 0, 0, 0, main.cuda_stream, main.cuda_stream,
                                                st.streams[1].cuda_stream), "coot_step_forward")
            torch.cuda.synchronize()
            outs[mode] = [v.cpu().numpy() for v in t]
    finally:
        cva.lib.check(lib.coot_set_option(b"pool_handover", 1), "pool_handover")
    import numpy as np
    for a, b_ in zip(outs[0], outs[1]):
        assert np.isfinite(a).all() and np.array_equal(a, b_)
    resh = outs[1][4]
    for v, cnt in enumerate(counts):  # padding rows are zero, item rows are the flat rows in order
        assert not resh[v, cnt:].any()
    flat = outs[1][0][B:]
    assert np.array_equal(np.concatenate([resh[v, :cnt] for v, cnt in enumerate(counts)]), flat)
    tr.close()
