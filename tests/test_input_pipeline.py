"""Input side: collation into one arena (coot_collate_level, host code of libcoot_hip.so) and the device staging
loader, against batches RetrievalDataset.collate_fn itself produced (tests/golden/collate.npz, oracle/gen_golden.py:
gen_collate).  Byte work: bit-exact."""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O

FIELDS = ("vid_feat", "vid_feat_mask", "vid_feat_len", "par_feat", "par_feat_mask", "par_feat_len", "clip_num", "clip_feat",
          "clip_feat_mask", "clip_feat_len", "sent_num", "sent_feat", "sent_feat_mask", "sent_feat_len")


def _points(seed, B, dv, dt, as_torch=False, **kw):
    import torch
    from coot_videotext_amd.dataset_retrieval import RetrievalDataPointTuple
    pts = []
    for d in O.make_datapoints(seed, B, dv, dt, **kw):
        conv = (lambda a: torch.from_numpy(a)) if as_torch else (lambda a: a)
        clips = [conv(c) for c in d["clip_feat_list"]]
        par = conv(d["par_feat"])
        sents, ptr = [], 0
        for n in d["sent_feat_len_list"]:
            sents.append(par[ptr:ptr + n]); ptr += n
        pts.append(RetrievalDataPointTuple(d["key"], d["key"], ["w"] * len(clips), conv(d["vid_feat"]), d["vid_feat"].shape[0], par,
                                           par.shape[0], len(clips), clips, [c.shape[0] for c in clips], len(clips), sents,
                                           d["sent_feat_len_list"]))
    return pts


@pytest.mark.parametrize("as_torch", [False, True])
def test_collate_matches_reference_batches(golden_dir, as_torch):
    from coot_videotext_amd.dataset_retrieval import collate_fn
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    for name in ("ragged", "single"):
        seed, B, dv, dt = (int(v) for v in g[name + "_args"])
        batch = collate_fn(_points(seed, B, dv, dt, as_torch=as_torch))
        for f in FIELDS:
            got, ref = getattr(batch, f).numpy(), g[f"{name}_{f}"]
            assert got.dtype == ref.dtype and got.shape == ref.shape, (name, f, got.dtype, ref.dtype, got.shape, ref.shape)
            assert np.array_equal(got, ref), (name, f)
        assert batch.max_clip_num == int(g[f"{name}_clip_num"].max()) and batch.max_sent_num == int(g[f"{name}_sent_num"].max())
        assert batch.key == [f"v_{seed}_{b}" for b in range(B)]


def test_collate_arena_reuse_threads_and_bf16():
    """One arena reused across batches of different shapes (stale bytes of the bigger batch must not leak into the padding of
    the smaller one), the threaded path (> 1 MB), and bf16 staging = torch's round-to-nearest-even cast of the fp32 batch,
    special values included."""
    import torch
    from coot_videotext_amd.dataset_retrieval import BatchArena, collate_fn
    arena = BatchArena(pin=False)
    big = _points(3, 16, 256, 64, max_frames=40, max_words=30)
    b1 = collate_fn(big, arena, threads=4)
    assert b1.clip_feat.numel() * 4 > (1 << 20)  # the clip level takes the threaded path
    ref1 = collate_fn(big, None, threads=1)
    for f in FIELDS:
        assert torch.equal(getattr(b1, f), getattr(ref1, f)), f
    snap = {f: getattr(ref1, f).clone() for f in FIELDS}
    small = _points(4, 3, 256, 64, max_frames=5, max_words=4)
    b2 = collate_fn(small, arena, threads=4)
    ref2 = collate_fn(small, None)
    for f in FIELDS:
        assert torch.equal(getattr(b2, f), getattr(ref2, f)), f
        assert torch.equal(getattr(ref1, f), snap[f])  # its own arena: untouched
    # padding really is zero / masked and lengths agree with the masks
    assert float(b2.clip_feat[b2.clip_feat_mask].abs().sum()) == 0.0
    assert torch.equal((~b2.clip_feat_mask).sum(1), b2.clip_feat_len) and torch.equal((~b2.sent_feat_mask).sum(1), b2.sent_feat_len)
    # bf16 staging
    special = np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, np.inf, -np.inf, np.nan, 1e-40, 3.3895314e38, -2.5, 65504.0],
                       dtype=np.float32)
    pts = _points(8, 4, 12, 12)
    pts[0].vid_feat[0, :12] = special
    bb = collate_fn(pts, None, bf16=True)
    bf = collate_fn(pts, None, bf16=False)
    for f in ("vid_feat", "clip_feat", "par_feat", "sent_feat"):
        got, want = getattr(bb, f), getattr(bf, f).to(torch.bfloat16)
        assert got.dtype == torch.bfloat16
        nan = torch.isnan(want)  # NaN payloads are not specified (torch's vectorised cast and the scalar one differ): NaN stays NaN
        assert torch.equal(torch.isnan(got), nan)
        assert np.array_equal(got.view(torch.int16)[~nan].numpy(), want.view(torch.int16)[~nan].numpy()), f
    assert torch.equal(bb.vid_feat_mask, bf.vid_feat_mask) and torch.equal(bb.sent_feat_len, bf.sent_feat_len)


def test_collate_level_rejects_bad_arguments():
    import ctypes as C
    import coot_videotext_amd as cva
    lib = cva.lib.load()
    src = np.zeros((3, 4), np.float32)
    dst = np.zeros((1, 2, 4), np.float32)
    seq = (C.c_void_p * 1)(src.ctypes.data)
    rows = (C.c_int64 * 1)(3)  # longer than max_rows = 2
    rc = lib.coot_collate_level(seq, rows, 1, 4, 2, 0, dst.ctypes.data, None, 1)
    assert rc != 0 and b"max_rows" in lib.coot_last_error()
    rows = (C.c_int64 * 1)(2)
    assert lib.coot_collate_level(seq, rows, 1, 0, 2, 0, dst.ctypes.data, None, 1) != 0
    assert lib.coot_collate_level(seq, rows, 1, 4, 2, 0, dst.ctypes.data, None, 1) == 0
    with pytest.raises(AssertionError):
        from coot_videotext_amd.dataset_retrieval import collate_fn
        collate_fn([])


@pytest.mark.gpu
def test_device_loader_stages_batches_and_feeds_training():
    """DeviceLoader: every batch arrives on the device bit-identical to the host collation, in order, with arenas rotating
    (more batches than slots); bf16 staging arrives as the fp32 widening of the bf16 cast; train_model consumes it."""
    import torch
    import coot_videotext_amd as cva
    from coot_videotext_amd.dataset_retrieval import DeviceLoader, collate_fn
    from tests import helpers as H
    assert torch.cuda.is_available()
    dims = (64, 48, 64, 4, 64, 128)
    lists = [_points(50 + i, 6, dims[0], dims[1], max_frames=12, max_words=9, as_torch=bool(i % 2)) for i in range(7)]
    loader = DeviceLoader(lists, depth=2)
    assert len(loader) == 7
    seen = 0
    keep = []
    for i, db in enumerate(loader):
        hb = collate_fn(lists[i])
        keep.append(db)  # views of rotating arenas: compare now, and hold on to them to prove later batches do not need them
        for f in FIELDS:
            assert getattr(db, f).is_cuda and torch.equal(getattr(db, f).cpu(), getattr(hb, f)), (i, f)
        assert db.key == hb.key and db.max_clip_num == hb.max_clip_num
        seen += 1
    assert seen == 7
    # the same with collation + copy enqueue on a loader thread, padded and packed at the source
    for packed in (False, True):
        got = 0
        for i, db in enumerate(DeviceLoader(lists, depth=2, background=True, packed=packed)):
            hb = collate_fn(lists[i], packed=packed)
            for f in (("vis_tokens", "txt_tokens", "cu_vis", "clip_feat_len") if packed else FIELDS):
                assert getattr(db, f).is_cuda and torch.equal(getattr(db, f).cpu(), getattr(hb, f)), (packed, i, f)
            assert db.key == hb.key
            got += 1
        assert got == 7
    for i, db in enumerate(DeviceLoader(lists[:3], depth=1, bf16=True)):
        hb = collate_fn(lists[i])
        assert db.clip_feat.dtype == torch.float32
        assert torch.equal(db.clip_feat.cpu(), hb.clip_feat.to(torch.bfloat16).float()) and torch.equal(db.sent_feat_mask.cpu(), hb.sent_feat_mask)
    # training straight from the loader (two epochs over 4 batches)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.01)
    cfg.raw["lr_scheduler"] = dict(name="none", warmup_type="none", warmup_epochs=0)
    cfg.train.num_epochs = 2
    for k, v in dict(val_freq=1, val_start=0, val_clips=False, val_clips_freq=1, det_best_field="val_score_at_1",
                     det_best_compare_mode="max", det_best_threshold_mode="rel", det_best_threshold_value=1e-4,
                     det_best_terminate_after=16).items():
        setattr(cfg.val, k, v)
    tr = cva.RetrievalTrainer(cfg, mgr)
    hist = tr.train_model(DeviceLoader(lists[:4], depth=2), DeviceLoader(lists[4:6], depth=1))
    assert len(hist["epoch"]) == 2 and all(np.isfinite(hist["train_loss"])) and hist["val"][1]["val_score_at_1"] >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("background", [False, True])
def test_device_loader_lookahead_keeps_the_previous_batch(background):
    """DeviceLoader(lookahead = 1): the consumer requests batch t + 1 BEFORE it works on batch t (train_model's lookahead: the step on
    batch t also normalises batch t + 1).  Batch t's arena must then survive one more request: device work enqueued on batch t AFTER
    batch t + 1 was requested still reads the right data, with more batches than slots; and train_model through such a loader (native
    steps with next_batch) gives the losses of the plain loader."""
    import torch
    import coot_videotext_amd as cva
    from coot_videotext_amd.dataset_retrieval import DeviceLoader, collate_fn
    from coot_videotext_amd.trainer_retrieval import _with_next
    from tests import helpers as H
    dims = (64, 48, 64, 4, 64, 128)
    lists = [_points(150 + i, 6, dims[0], dims[1], max_frames=12, max_words=9) for i in range(9)]
    sums = []
    for cur, nxt in _with_next(DeviceLoader(lists, depth=1, lookahead=1, background=background)):
        # (nxt has been requested: a loader without lookahead would have released cur's arena before this line)
        sums.append((cur.vid_feat.double().sum() + cur.clip_feat.double().sum() + cur.sent_feat.double().sum()).clone())
    torch.cuda.synchronize()
    assert len(sums) == 9
    for i, s_ in enumerate(sums):
        hb = collate_fn(lists[i])
        want = float(hb.vid_feat.double().sum() + hb.clip_feat.double().sum() + hb.sent_feat.double().sum())
        assert abs(float(s_) - want) <= 1e-9 * max(1.0, abs(want)), i
    # train_model: one epoch over 5 batches, lookahead loader against the plain one
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    hist = []
    for la in (0, 1):
        cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.0)
        cfg.optimizer.adam_eps = 1e-3
        cfg.raw["lr_scheduler"] = dict(name="none", warmup_type="none", warmup_epochs=0)
        cfg.train.num_epochs = 1
        for k, v in dict(val_freq=1, val_start=0, val_clips=False, val_clips_freq=1, det_best_field="val_score_at_1",
                         det_best_compare_mode="max", det_best_threshold_mode="rel", det_best_threshold_value=1e-4,
                         det_best_terminate_after=16).items():
            setattr(cfg.val, k, v)
        tr = cva.RetrievalTrainer(cfg, mgr)
        tr.lookahead_min_stage_bytes = 0  # (the trainer skips the lookahead for inputs as small as these)
        h = tr.train_model(DeviceLoader(lists[:5], depth=1, lookahead=la, background=background), DeviceLoader(lists[5:7], depth=1))
        assert (getattr(tr._native, "stages", None) is not None) == bool(la)
        hist.append(h["train_loss"][0])
    assert abs(hist[0] - hist[1]) <= 1e-3 * abs(hist[0]), hist  # (run-to-run drift of five updates: fp32 atomics order; a stale batch moves it by O(0.1))


def _collate_numpy(pts):
    """Plain restatement of RetrievalDataset.collate_fn (coot/dataset_retrieval.py:335-463) with numpy loops: the checker for
    the randomised shapes below (the reference-generated fixture pins two batches; this pins the layout rule for many)."""
    B = len(pts)
    dv, dt = pts[0].vid_feat.shape[-1], pts[0].par_feat.shape[-1]
    vl = [p.vid_feat_len for p in pts]; pl = [p.par_feat_len for p in pts]
    vid = np.zeros((B, max(vl), dv), np.float32); vmask = np.ones((B, max(vl)), bool)
    par = np.zeros((B, max(pl), dt), np.float32); pmask = np.ones((B, max(pl)), bool)
    for b, p in enumerate(pts):
        vid[b, :vl[b]] = p.vid_feat; vmask[b, :vl[b]] = False
        par[b, :pl[b]] = p.par_feat; pmask[b, :pl[b]] = False
    cl = [c.shape[0] for p in pts for c in p.clip_feat_list]
    clip = np.zeros((len(cl), max(cl), dv), np.float32); cmask = np.ones((len(cl), max(cl)), bool)
    i = 0
    for p in pts:
        for c in p.clip_feat_list:
            clip[i, :c.shape[0]] = c; cmask[i, :c.shape[0]] = False; i += 1
    sl = [n for p in pts for n in p.sent_feat_len_list]
    sent = np.zeros((len(sl), max(sl), dt), np.float32); smask = np.ones((len(sl), max(sl)), bool)
    i = 0
    for b, p in enumerate(pts):
        ptr = 0
        for n in p.sent_feat_len_list:
            sent[i, :n] = par[b, ptr:ptr + n]; smask[i, :n] = False; i += 1; ptr += n
    return dict(vid_feat=vid, vid_feat_mask=vmask, vid_feat_len=np.array(vl), par_feat=par, par_feat_mask=pmask, par_feat_len=np.array(pl),
                clip_num=np.array([p.clip_num for p in pts]), clip_feat=clip, clip_feat_mask=cmask, clip_feat_len=np.array(cl),
                sent_num=np.array([p.sent_num for p in pts]), sent_feat=sent, sent_feat_mask=smask, sent_feat_len=np.array(sl))


def test_collate_random_shapes_match_layout_rule():
    """40 random batches (1-9 videos, 1-6 clips, 1-23 frames / 1-11 words, odd feature widths) through ONE reused arena with
    1-5 threads: every field equals the loop restatement of the reference's collate rule."""
    from coot_videotext_amd.dataset_retrieval import BatchArena, collate_fn
    rs = np.random.RandomState(77)
    arena = BatchArena(pin=False)
    for trial in range(40):
        B, dv, dt = int(rs.randint(1, 10)), int(rs.choice([4, 12, 20, 36])), int(rs.choice([4, 8, 28]))
        pts = _points(1000 + trial, B, dv, dt, max_frames=int(rs.randint(1, 24)), max_words=int(rs.randint(1, 12)),
                      max_clips=int(rs.randint(1, 7)))
        want = _collate_numpy(pts)
        got = collate_fn(pts, arena, threads=int(rs.randint(1, 6)))
        for f in FIELDS:
            g = getattr(got, f).numpy()
            assert g.shape == want[f].shape and np.array_equal(g, want[f]), (trial, f)


# ---- packed at the source: coot_collate_packed / collate_fn(packed=True) / unpack_batch ---------------------------

def test_packed_collation_unpacks_to_the_reference_batch(golden_dir):
    """collate_fn(packed=True) writes no padding row; unpack_batch() of it is the batch RetrievalDataset.collate_fn itself produced
    (tests/golden/collate.npz), bit for bit — features, masks, lengths."""
    import torch
    from coot_videotext_amd.dataset_retrieval import collate_fn, unpack_batch
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    for name in ("ragged", "single"):
        seed, B, dv, dt = (int(v) for v in g[name + "_args"])
        pb = collate_fn(_points(seed, B, dv, dt), packed=True)
        Nc = int(g[f"{name}_clip_num"].sum())
        assert pb.vis_tokens.shape == (int(g[f"{name}_vid_feat_len"].sum() + g[f"{name}_clip_feat_len"].sum()), dv)
        assert pb.txt_tokens.shape == (int(g[f"{name}_par_feat_len"].sum() + g[f"{name}_sent_feat_len"].sum()), dt)
        assert pb.tok_vis == pb.vis_tokens.shape[0] and pb.tok_txt == pb.txt_tokens.shape[0]
        lens_v = np.concatenate([g[f"{name}_vid_feat_len"], g[f"{name}_clip_feat_len"]])
        assert pb.cu_vis.dtype == torch.int32 and np.array_equal(pb.cu_vis.numpy(), np.concatenate([[0], np.cumsum(lens_v)]))
        assert pb.cu_vis.numel() == B + Nc + 1 and pb.max_lens == tuple(int(g[f"{name}_{k}"].shape[1]) for k in ("vid_feat", "clip_feat", "par_feat", "sent_feat"))
        batch = unpack_batch(pb)
        for f in FIELDS:
            got, ref = getattr(batch, f).numpy(), g[f"{name}_{f}"]
            assert got.dtype == ref.dtype and got.shape == ref.shape and np.array_equal(got, ref), (name, f)


def test_packed_collation_threads_bf16_and_bad_arguments():
    """The threaded path (runs of sequences with equal row counts per thread) equals the serial one; bf16 rows are torch's
    round-to-nearest-even cast of the fp32 rows; an arena reused for a smaller batch leaves nothing behind that is read."""
    import ctypes as C
    import torch
    import coot_videotext_amd as cva
    from coot_videotext_amd.dataset_retrieval import BatchArena, collate_fn, unpack_batch
    pts = _points(3, 16, 256, 64, max_frames=40, max_words=30)
    a = collate_fn(pts, None, threads=1, packed=True)
    arena = BatchArena(pin=False)
    b = collate_fn(pts, arena, threads=5, packed=True)
    assert a.vis_tokens.numel() * 4 > (1 << 20)
    for f in ("vis_tokens", "txt_tokens", "cu_vis", "cu_txt", "clip_feat_len", "sent_num"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    h = collate_fn(pts, None, bf16=True, threads=3, packed=True)
    assert h.vis_tokens.dtype == torch.bfloat16 and torch.equal(h.vis_tokens, a.vis_tokens.to(torch.bfloat16)) and torch.equal(h.txt_tokens, a.txt_tokens.to(torch.bfloat16))
    small = _points(4, 3, 256, 64, max_frames=9, max_words=7)
    s1, s2 = collate_fn(small, arena, packed=True), collate_fn(small, None, packed=True)
    ub1, ub2 = unpack_batch(s1), unpack_batch(s2)
    for f in FIELDS:
        assert torch.equal(getattr(ub1, f), getattr(ub2, f)), f
    lib = cva.lib.load()
    cu = (C.c_int32 * 3)()
    rows = (C.c_int64 * 2)(1, -1)
    x = np.zeros((4, 8), np.float32)
    seq = (C.c_void_p * 2)(x.ctypes.data, x.ctypes.data)
    out = np.zeros((8, 8), np.float32)
    assert lib.coot_collate_packed(seq, rows, 2, 8, 0, out.ctypes.data, cu, 1) != 0 and b"rows" in lib.coot_last_error()
    assert lib.coot_collate_packed(seq, rows, 2, 8, 0, out.ctypes.data, None, 1) != 0


@pytest.mark.gpu
def test_packed_source_step_equals_the_padded_batch_step():
    """coot_train_step on a batch packed at the source (COOT_SOURCE_PACKED_F32: the input LayerNorm reads the packed rows in place)
    gives the losses and gradients of the same batch in the reference's padded layout with cu_seqlens (the gather path) — the
    same rows, the same arithmetic: identical up to the summation order of atomics; bf16 rows (COOT_SOURCE_PACKED_BF16) within
    the bf16 rounding of the features.  Through DeviceLoader(packed=True), as a training loop would."""
    import torch
    import coot_videotext_amd as cva
    from coot_videotext_amd.dataset_retrieval import DeviceLoader, collate_fn, unpack_batch
    from tests import helpers as H
    dims = (256, 128, 384, 8, 384, 768)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 3 + i, scale=0.05) for i in range(4)]
    pts = _points(9, 12, dims[0], dims[1], max_frames=80, max_words=40)   # ~2 000 valid frames: the fused packed path on the video side
    res = {}
    for mode in ("padded+cu", "packed f32", "packed bf16"):
        cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.01)
        mgr.set_all_models_train()
        tr = cva.RetrievalTrainer(cfg, mgr)
        if mode == "padded+cu":
            batch = collate_fn(pts)
            batch.to_cuda()
            cva.attach_packed_index(batch)
        else:
            batch = next(iter(DeviceLoader([pts], depth=1, bf16=(mode == "packed bf16"), packed=True)))
            assert isinstance(batch, cva.RetrievalPackedBatchTuple) and batch.vis_tokens.is_cuda
        B = len(pts)
        idx = torch.zeros(2 * B, dtype=torch.int64, device="cuda")
        losses = [float(v) for v in tr.train_step_native(batch, do_optimizer=False, cc_indices=idx)]
        torch.cuda.synchronize()
        res[mode] = (losses, [n._grad_flat.detach().cpu().numpy().copy() for n in mgr.model_dict.values()])
        if mode == "packed f32":  # the packed batch is the padded one without its padding
            ub, hb = unpack_batch(batch), collate_fn(pts)
            for f in FIELDS:
                assert torch.equal(getattr(ub, f).cpu(), getattr(hb, f)), f
    (l0, g0), (l1, g1), (l2, g2) = res["padded+cu"], res["packed f32"], res["packed bf16"]
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
    for a, b_ in zip(g0, g1):
        assert float(np.abs(a - b_).max()) <= 2e-4 * float(np.abs(a).max())
    assert np.allclose(l0, l2, rtol=5e-3), (l0, l2)
    for a, b_ in zip(g0, g2):
        assert H.cosine_flat(a, b_) > 0.999


@pytest.mark.gpu
def test_packed_source_fallback_copy_outlives_the_launch_with_lookahead():
    """A packed-source batch outside the packed-row kernels' domain (here: fewer than 1 024 tokens) is unpacked on the device by the
    trainer, and the step's raw pointers refer to that temporary padded copy.  The copy must stay referenced until the step has been
    enqueued: with a lookahead, unpacking the NEXT batch allocates tensors of the same sizes in between, and a copy released too
    early is overwritten by them before the launch (the step then trains on the next batch's features with this batch's lengths —
    a loss off by O(0.1)).  Steps on packed batches with next_batch == steps on explicitly unpacked, explicitly held batches."""
    import torch
    import coot_videotext_amd as cva
    from coot_videotext_amd.dataset_retrieval import DeviceLoader, unpack_batch
    from tests import helpers as H
    dims = (64, 48, 64, 4, 64, 128)
    lists = [_points(250 + i, 6, dims[0], dims[1], max_frames=12, max_words=9) for i in range(5)]
    packed = list(DeviceLoader(lists, depth=5, packed=True))
    assert all(isinstance(b, cva.model_retrieval.RetrievalPackedBatchTuple) and min(b.tok_vis, b.tok_txt) < 1024 for b in packed)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    res = []
    for mode in ("packed+lookahead", "unpacked"):
        cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
        cfg_x.optimizer.adam_eps = 1e-3  # (see tests/test_gpu_path.py::test_deferred_text_join_gives_the_same_training_trajectory)
        mgr.set_all_models_train()
        tr = cva.RetrievalTrainer(cfg_x, mgr)
        tr.lookahead_min_stage_bytes = 0
        held = [unpack_batch(b) for b in packed] if mode == "unpacked" else None
        losses = []
        for it in range(5):
            if held is None:
                nxt = packed[it + 1] if it + 1 < 5 else None
                out = tr.train_step_native(packed[it], seed=300 + it, next_batch=nxt)
            else:
                out = tr.train_step_native(held[it], seed=300 + it)
            losses.append([float(v) for v in out])
        torch.cuda.synchronize()
        res.append(losses)
    la, lb = res
    assert np.allclose(la[:2], lb[:2], rtol=1e-4, atol=1e-7), (la, lb)
    assert np.allclose(la, lb, rtol=1e-3, atol=1e-6), (la, lb)
