"""CPU-only checks: the C-ABI library loads and exports every symbol include/coot_hip.h declares (no compute
calls), host-side logic (config schema, parameter layout, state-dict names, metrics), product path fails loudly
without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cva():
    import coot_videotext_amd as m
    return m


def test_library_exports_every_declared_symbol(cva):
    hdr = open(os.path.join(ROOT, "include", "coot_hip.h")).read()
    declared = set(re.findall(r"\b(coot_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"coot_stream_t"}
    lib = cva.lib.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(cva.lib.EXPORTS) <= declared
    assert lib.coot_version() == cva.lib.ABI_VERSION == int(re.search(r"#define COOT_ABI_VERSION (\d+)", hdr).group(1))
    # ... and nothing else leaves the shared object (-fvisibility=hidden + csrc/exports.map): no C++ internals, no kernel handles
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", cva.lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared)[:5], sorted(declared - exported)[:5])
    # the IEEE-half build of the same sources (csrc/build.sh: libcoot_hip_f16.so) has the same dynamic symbol table and ABI version
    f16 = os.path.join(os.path.dirname(cva.lib.LIB_PATH), "libcoot_hip_f16.so" if cva.lib.OPERAND_ENV == "bf16" else "libcoot_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", f16], capture_output=True, text=True, check=True).stdout
    assert {ln.split()[-1] for ln in out.splitlines() if ln.strip()} == declared
    import ctypes
    assert ctypes.CDLL(f16).coot_version() == cva.lib.ABI_VERSION


def _header_prototypes():
    """name -> number of parameters, parsed from include/coot_hip.h (declarations end in ');' and contain no function pointers)."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "coot_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(coot_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def test_ctypes_argtypes_match_the_header(cva):
    """Every function lib.py gives argtypes takes exactly as many arguments as include/coot_hip.h declares (a binding that lags a
    signature change would shift every later argument: e.g. a stream passed where seed_dev is expected); the ctypes stub printed
    in INTEGRATION.md is held to the same count."""
    import re
    lib = cva.lib.load()
    protos = _header_prototypes()
    assert len(protos) >= 55
    checked = 0
    for name, n in protos.items():
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue
        assert len(fn.argtypes) == n, (name, len(fn.argtypes), n)
        checked += 1
    assert checked >= 50, checked
    # calls of the form lib.coot_xxx(a, b, ...) inside its python blocks
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    doc = re.sub(r"#[^\n]*", "", doc)  # python comments (they may hold commas)
    calls = 0
    for m in re.finditer(r"lib\.(coot_[a-z0-9_]+)\(", doc):
        name, i, depth, nargs, seen = m.group(1), m.end(), 1, 0, False
        while depth and i < len(doc):
            ch = doc[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                nargs += 1
            elif not ch.isspace() and depth >= 1:
                seen = True
            i += 1
        if depth == 0 and name in protos and not doc[m.end():i - 1].strip().startswith("..."):
            got = nargs + 1 if seen else 0
            assert got == protos[name], ("INTEGRATION.md", name, got, protos[name])
            calls += 1
    assert calls >= 1


def test_dropout_mask_restatement_matches_the_library(cva):
    """oracle/dropout_masks.py (the masks injected into the reference for train-mode parity) against the library's own generator
    evaluated on the host by the same functions the kernels use (coot_debug_dropout_scales / coot_debug_attn_dropout_scales):
    hash, key derivation, pair halves, quantisation of p, 1 / keep scale — bit-exact."""
    import ctypes as C
    from oracle import dropout_masks as DM
    lib = cva.lib.load()
    lib.coot_debug_dropout_scales.argtypes = [C.c_uint64, C.c_uint, C.c_uint64, C.c_int64, C.c_float, C.c_void_p]
    lib.coot_debug_attn_dropout_scales.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_int, C.c_float, C.c_void_p]
    rs = np.random.RandomState(0)
    for seed, site, p in ((0, 1, 0.1), (20250926, 16 * 15 + 7, 0.025), (2 ** 63 + 12345, 16 * 8 + 3, 0.5), (77001 + 1022, 4, 1e-6),
                          ((1 << 64) - 1, 2, 0.9999)):
        for idx0 in (0, 1, 2 ** 32 - 3, int(rs.randint(0, 2 ** 31)) * 5):
            n = 4099
            out = np.empty(n, dtype=np.float32)
            cva.lib.check(lib.coot_debug_dropout_scales(seed, site, idx0, n, p, out.ctypes.data), "debug_dropout_scales")
            want = DM.scales_from_index(DM.drop_key(seed, site), np.arange(idx0, idx0 + n, dtype=np.uint64), p)
            assert np.array_equal(out, want), (seed, site, p, idx0)
            if idx0 == 0 and 0.01 < p < 0.9:
                assert abs((out == 0).mean() - p) < 0.03  # the masks drop about p of the elements, the others carry 1 / keep
                assert np.allclose(out[out != 0], 1.0 / (1.0 - DM.quantise_p(p)[0] / 65536.0))
        for row32, Lk in ((0, 80), (12345, 17), (2 ** 32 - 1, 64), (640 * 8 * 80 + 7, 1)):
            out = np.empty(Lk, dtype=np.float32)
            cva.lib.check(lib.coot_debug_attn_dropout_scales(seed, site, row32, Lk, p, out.ctypes.data), "debug_attn_dropout_scales")
            want = DM.scales_attn(DM.drop_key(seed, site), np.full(Lk, row32, np.uint64), np.arange(Lk), (Lk + 1) >> 1, p)
            assert np.array_equal(out, want), (seed, site, p, row32, Lk)
    # layouts: for full-length sequences the packed and the padded layout draw the same masks (cu[n] = n L)
    N, L, H = 3, 6, 2
    pad = DM.CallLayout(0, N, L, 0)
    pk = DM.CallLayout(0, N, L, 0, lens=[L] * N, cu=np.arange(N) * L)
    assert np.array_equal(DM.mask_rows(5, 3, pad, 8, 0.3), DM.mask_rows(5, 3, pk, 8, 0.3))
    assert np.array_equal(DM.mask_attention(5, 1, pad, H, L, L, 0.3), DM.mask_attention(5, 1, pk, H, L, L, 0.3))
    # second segment of a call: element-wise sites continue the row count, attention / pooling weights restart under another seed
    seg1 = DM.CallLayout(1, 2, 4, N * L)
    both = DM.mask_rows(5, 3, DM.CallLayout(0, 1, N * L + 8, 0), 8, 0.3)[0]
    assert np.array_equal(DM.mask_rows(5, 3, seg1, 8, 0.3).reshape(8, 8), both[N * L:])
    assert not np.array_equal(DM.mask_attention(5, 1, seg1, H, 4, 4, 0.3), DM.mask_attention(5, 1, DM.CallLayout(0, 2, 4, 0), H, 4, 4, 0.3))
    assert DM.step_net_seeds(100) == [100, 111, 1122, 1133]


def test_param_layout_matches_reference_state_dict(cva, golden_dir):
    g = np.load(os.path.join(golden_dir, "full_anet.npz"))
    cfgs = H.full_cfgs(2048, 1536, 384, 8, 384, 768)
    ref_counts = {"net_video_local": 2123009 - 1, "net_video_global": 1777920, "net_text_local": 1925377 - 1,
                  "net_text_global": 1777920}  # SURVEY 8a row a2 (minus the non-trainable genpool_one)
    for key, oc in zip(H.NET_KEYS, cfgs):
        tc = cva.TransformerConfig(H.ocfg_to_dict(oc), oc.input_dim)
        total, table = cva.lib.param_table(tc.to_c())
        assert total == ref_counts[key]
        shapes = dict(O.param_shapes(oc))
        assert {n: s for n, _, s in table} == shapes
        # names are exactly the reference's named_parameters (as recorded in the golden fixture)
        ref_names = {k.split(":", 2)[2] for k in g.files if k.startswith(f"gnorm:{key}:")}
        assert {n for n, _, _ in table} == ref_names
        # non-overlapping, dense
        spans = sorted((o, o + int(np.prod(s))) for _, o, s in table)
        assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == total


def test_module_state_dict_roundtrip_and_flat_views(cva):
    import torch
    oc = O.NetConfig(input_dim=40, hidden_dim=64, num_heads=4, ff_dim=64, pool_hidden=128)
    P = O.make_params(oc, 1)
    net = H.make_hip_net(oc, P, device="cpu")
    sd = net.state_dict()
    assert "embedding.pe" in sd and "pooler.pools.0.genpool_one" in sd
    assert np.allclose(sd["embedding.pe"].numpy(), O.sincos_pe(1000, 64), atol=1e-6)
    for n, _, _ in net.table:
        assert np.allclose(sd[n].numpy(), P[n].astype(np.float32))
    # parameters are views of one flat arena; in-place optimizer-style updates are visible in it
    with torch.no_grad():
        net._params[3].add_(1.0)
    n, off, shape = net.table[3]
    assert torch.equal(net._flat[off:off + net._params[3].numel()].view(shape), net._params[3].detach())
    net2 = H.make_hip_net(oc, O.make_params(oc, 2), device="cpu")
    net2.load_state_dict(net.state_dict())
    assert torch.equal(net2._flat, net._flat)


def test_reference_init_statistics(cva):
    tc = cva.TransformerConfig(H.ocfg_to_dict(O.NetConfig(input_dim=64, hidden_dim=64, num_heads=4, ff_dim=64, pool_hidden=128)), 64)
    import torch
    torch.manual_seed(0)
    net = cva.TransformerHip(tc)
    sd = net.state_dict()
    w = sd["tf.encoder_layers.0.self_attention_layer.sublayer.query_projection.weight"]
    assert float(w.abs().max()) <= 0.02 + 1e-6 and 0.007 < float(w.std()) < 0.011  # truncnorm(0.01, +-2 sigma)
    b = sd["tf.encoder_layers.0.self_attention_layer.sublayer.query_projection.bias"]
    assert float(b.abs().max()) > 0  # biases are initialised too (initialization.py:100-105)
    assert torch.equal(sd["norm_input.gain"], torch.ones(64)) and torch.equal(sd["norm_input.bias"], torch.zeros(64))


def test_config_schema_and_same_as(cva):
    cfg = cva.load_named_config("anet_coot")
    loc, glob = cfg.model_cfgs["net_video_local"], cfg.model_cfgs["net_text_global"]
    assert (loc.input_dim, loc.hidden_dim, loc.num_heads, loc.ff_dim, loc.pool_hidden, loc.pool_heads) == (2048, 384, 8, 384, 768, 2)
    assert cfg.model_cfgs["net_text_local"].input_dim == 1536
    assert glob.use_context and not glob.use_input_fc and glob.pooler == "avg_special" and glob.input_dim == 384
    assert abs(loc.dropout - 0.025) < 1e-9 and cfg.train.loss_cycle_cons == 0.01
    assert cva.load_named_config("yc2_100m_coot").model_cfgs["net_video_local"].input_dim == 512
    assert cva.load_named_config("yc2_2d3d_coot").model_cfgs["net_video_local"].input_dim == 4096
    raw = dict(a=dict(x=1, y=dict(z=2)), b=dict(same_as="a", x=5), c=dict(same_as="b", y=dict(z=7)))
    out = cva.config.resolve_same_as(raw)
    assert out["b"] == dict(x=5, y=dict(z=2)) and out["c"] == dict(x=5, y=dict(z=7))
    mgr = cva.RetrievalModelManager(cfg)
    params, names, flat = mgr.get_all_params()
    assert len(params) == len(names) == len(flat) and sum(p.numel() for p in flat) == 7604224
    assert all((p["decay_mult"] == 0.0) == ("bias" in n) for p, n in zip(params, names))


def test_retrieval_metrics_match_reference(cva, golden_dir):
    g = np.load(os.path.join(golden_dir, "retrieval_metrics.npz"))
    for i in range(3):
        res, _top1, ranks = cva.compute_retrieval_cosine(g[f"d{i}"])
        assert (ranks == g[f"ranks{i}"]).all()
        assert np.allclose([res[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")], g[f"res{i}"])


def test_product_path_has_no_cpu_fallback(cva):
    import torch
    oc = O.NetConfig(input_dim=40, hidden_dim=64, num_heads=4, ff_dim=64, pool_hidden=128)
    net = H.make_hip_net(oc, O.make_params(oc, 1), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(2, 3, 40), None, torch.tensor([3, 2]), None)
    # nothing under the package imports, links or executes the oracle
    pkg = os.path.join(ROOT, "coot-videotext_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle|oracle\.coot_oracle\s*import|#include\s+\"[^\"]*oracle", re.M)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                assert not pat.search(open(os.path.join(dp, f)).read()), (dp, f)


def test_unsupported_options_raise(cva):
    d = H.ocfg_to_dict(O.NetConfig(input_dim=64, hidden_dim=64, num_heads=4, ff_dim=64))
    d["add_local_cls_token"] = True
    with pytest.raises(NotImplementedError):
        cva.TransformerConfig(d, 64)
    bad = cva.TransformerConfig(H.ocfg_to_dict(O.NetConfig(input_dim=64, hidden_dim=60, num_heads=4, ff_dim=64)), 64)
    with pytest.raises(RuntimeError, match="d_head"):
        cva.lib.param_table(bad.to_c())


def test_radam_optimizer_matches_reference_trajectory(cva):
    """The package's torch RAdam (autograd path of the yc2 configs: optimizer.name = radam) against the trajectory the
    reference's in-file class produced (tests/golden/radam.npz, oracle/gen_golden.py) — two parameter groups, both
    degenerated_to_sgd settings, all three phases of the rule."""
    import torch
    from coot_videotext_amd.trainer_retrieval import RAdam, make_optimizer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "radam.npz"))
    lr, b1, b2, eps, wd = (float(g[k]) for k in ("lr", "beta1", "beta2", "eps", "wd"))
    for degen in (0, 1):
        pa = torch.nn.Parameter(torch.from_numpy(g["p0"][:200].copy()))
        pb = torch.nn.Parameter(torch.from_numpy(g["p0"][200:].copy()))
        opt = RAdam([dict(params=[pa], weight_decay=wd), dict(params=[pb], weight_decay=0.0)], lr=lr, betas=(b1, b2), eps=eps,
                    degenerated_to_sgd=bool(degen))
        for s_ in range(len(g["grads"])):
            pa.grad = torch.from_numpy(g["grads"][s_][:200].copy()); pb.grad = torch.from_numpy(g["grads"][s_][200:].copy())
            opt.step()
            got = np.concatenate([pa.detach().numpy(), pb.detach().numpy()])
            ref = g[f"traj_degen{degen}"][s_]
            assert np.abs(got - ref).max() <= 2e-7 + 2e-6 * np.abs(ref).max(), (degen, s_, np.abs(got - ref).max())
    # make_optimizer dispatch (nntrainer/optimization.py:45-74)
    sec = type("S", (), dict(name="radam", lr=1e-3, weight_decay=1e-2, momentum=0.56, adam_beta2=0.98, adam_eps=1e-8, radam_degentosgd=False))
    p = torch.nn.Parameter(torch.zeros(3))
    o = make_optimizer(sec, [dict(params=p, decay_mult=1.0)])
    assert isinstance(o, RAdam) and o.degenerated_to_sgd is False and o.param_groups[0]["weight_decay"] == 1e-2
    sec.name = "sgd"
    with pytest.raises(NotImplementedError):
        make_optimizer(sec, [dict(params=p, decay_mult=1.0)])


def test_lr_schedule_matches_reference(cva, golden_dir):
    """lr_scheduler.py against nntrainer/lr_scheduler.py driven the way the trainer drives it (tests/golden/lr_schedule.npz,
    oracle/gen_golden.py: gen_lr_schedule): the learning rate of both parameter groups and current_lr after EVERY step() /
    step_epoch() call of the shipped ANet / YC2 schedules, per-step warmup, no warmup, a minimum LR factor, the constant
    schedule.  Host arithmetic in double precision: exact to rounding (1e-15 relative)."""
    import json
    import torch
    from coot_videotext_amd import lr_scheduler as lrs
    g = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    for name, sc, spe, epochs in json.loads(str(g["cases_json"])):
        pa, pb = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
        opt = torch.optim.Adam([dict(params=[pa], lr=1e-3), dict(params=[pb], lr=2.5e-4)], lr=1e-3)
        sched = lrs.make_lr_scheduler(opt, lrs.SchedulerConfig(sc), 1e-3, epochs, spe)
        flags = g[name + "_flags"]
        rows = [[opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr]]
        for e in range(epochs):
            for _ in range(spe):
                sched.step()
                rows.append([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr])
            sched.step_epoch(bool(flags[0, e]), bool(flags[1, e]))
            rows.append([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr])
        got, ref = np.array(rows), g[name]
        assert got.shape == ref.shape
        assert np.allclose(got, ref, rtol=1e-14, atol=0), (name, np.abs(got / np.maximum(ref, 1e-300) - 1).max())
        assert getattr(sched, "reduce_steps", 0) == int(g[name + "_reductions"])
    # the schedule reduced at least twice in the plateau cases (the fixture exercises what it claims to)
    assert int(g["anet_reductions"]) >= 2 and int(g["stepwarm_reductions"]) >= 3
    # a step the trainer did not announce is an error (nntrainer/lr_scheduler.py:213-223), and so is an unknown schedule name
    with pytest.raises(AssertionError):
        for _ in range(spe + 1):
            sched.step()
    with pytest.raises(ValueError):
        lrs.make_lr_scheduler(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]), lrs.SchedulerConfig(
            dict(name="cosine", warmup_type="none", warmup_epochs=0)), 1e-3, 1, 1)
    # state round trip
    sd = sched.state_dict()
    assert "optimizer" not in sd and sd["current_epoch"] == sched.current_epoch


def test_train_model_epoch_loop_early_stop_and_schedule(cva):
    """RetrievalTrainer.train_model (coot/trainer_retrieval.py:235-310 + the nntrainer/trainer_base.py hooks :285-353,
    :461-499) with scripted steps and validation results: new-best rule with the relative threshold, plateau reductions of the
    ANet schedule, early stop after det_best_terminate_after epochs, the learning rate every step sees."""
    import torch
    from coot_videotext_amd.trainer_retrieval import RetrievalTrainer
    cfg = cva.load_named_config("anet_coot")
    tr = object.__new__(RetrievalTrainer)
    tr.cfg = cfg
    p = torch.nn.Parameter(torch.zeros(2))
    tr.optimizer = torch.optim.Adam([p], lr=float(cfg.optimizer.lr))
    tr.lr_scheduler, tr.current_epoch, tr.det_best_field_best, tr.det_best_field_current = None, 0, None, 0.0
    tr.infos_val_epochs, tr.infos_val_is_good = [], []
    tr.model_mgr = type("M", (), dict(set_all_models_train=lambda self: None))()
    seen_lr = []
    # validation score: rises for 5 epochs, then a gain below the relative threshold (1e-4), then flat
    scores = [0.10, 0.12, 0.15, 0.20, 0.30] + [0.30 * (1 + 5e-5)] + [0.30] * 100

    def train_step(batch):
        seen_lr.append(tr.optimizer.param_groups[0]["lr"])
        return (torch.tensor(1.0), torch.tensor(0.9), torch.tensor(0.1))

    def validate_epoch(loader, val_clips=True):
        assert val_clips is False  # anet: val_clips false
        return {"val_score_at_1": scores[tr.current_epoch], "loss": 1.0}

    tr.train_step, tr.validate_epoch = train_step, validate_epoch
    hist = tr.train_model([0, 1, 2], [0], native=False)
    # last new best: epoch 4 -> epochs 0 .. 20 run (bad epochs = 20 - 4 = 16 stops epoch 21)
    assert hist["epoch"] == list(range(21)) and tr.current_epoch == 21
    assert tr.infos_val_is_good == [True] * 5 + [False] * 16 and abs(tr.det_best_field_best - 0.30) < 1e-12
    base = float(cfg.optimizer.lr)
    # warmup by epoch: epochs 0, 1, 2 train at 1/3, 2/3, 3/3 of the base LR
    assert np.allclose(seen_lr[:9], [base / 3] * 3 + [base * 2 / 3] * 3 + [base] * 3, rtol=1e-12)
    # plateau: bad epochs 5, 6, 7 (> patience 2) -> x0.1 from epoch 8; cooldown 8-10, bad 11-13 -> x0.01 from 14; 17-19 -> x0.001 from 20
    per_epoch = np.array(seen_lr).reshape(21, 3)[:, 0]
    want = [base / 3, base * 2 / 3] + [base] * 6 + [base * 0.1] * 6 + [base * 0.1 ** 2] * 6 + [base * 0.1 ** 3]
    assert np.allclose(per_epoch, want, rtol=1e-12), per_epoch
    assert np.allclose(hist["train_loss"], 1.0)


def test_embedding_export_hdf5_branch_with_a_recording_h5py(tmp_path, monkeypatch):
    """save_embeddings' HDF5 branch (coot/trainer_retrieval.py:404-415: ``with h5py.File(filename, mode="w") as h5: h5[name] =
    array``).  h5py is not installed in this image, so the branch runs against a recording stand-in with h5py's call surface
    (File(path, mode=...) as a context manager, item assignment): what is checked is the call pattern, the dataset names the
    consumers read (mart/recursive_caption_dataset.py:159-201, test_embeddings_retrieval.py:21-38) and the value types h5py
    accepts (numeric ndarrays, a list of str for ``key``) — the file format itself is h5py's business."""
    import sys
    import types
    from coot_videotext_amd.trainer_retrieval import save_embeddings
    written = {}

    class _File:
        def __init__(self, path, mode="r"):
            assert mode == "w"
            self.path, self.d = path, {}

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            written[self.path] = self.d
            open(self.path, "wb").close()
            return False

        def __setitem__(self, k, v):
            assert isinstance(k, str) and k not in self.d
            if isinstance(v, (list, tuple)):   # h5py stores a list of str as a variable-length string dataset
                assert all(isinstance(x, str) for x in v)
            else:
                assert isinstance(v, np.ndarray) and v.dtype.kind in "fiu", (k, type(v))
            self.d[k] = v

    fake = types.ModuleType("h5py")
    fake.File = _File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    names = ["vid_emb", "par_emb", "clip_emb", "sent_emb", "vid_context", "par_context"]
    rs = np.random.RandomState(0)
    emb = {"clip_num": np.array([2, 1, 3]), "sent_num": np.array([2, 1, 3]), "key": ["a", "b", "c"]}
    for n in names:
        raw = rs.randn(6 if n in ("clip_emb", "sent_emb") else 3, 8).astype(np.float32)
        emb[n + "_before_norm"] = raw
        emb[n] = raw / np.sqrt((raw * raw).sum(-1, keepdims=True))
    fn = save_embeddings(emb, str(tmp_path / "embeddings_7.h5"))
    assert fn.endswith("embeddings_7.h5") and os.path.exists(fn)
    d = written[fn]
    assert set(d) == {"clip_num", "sent_num", "key"} | set(names) | {n + "_before_norm" for n in names}
    assert d["key"] == ["a", "b", "c"] and np.array_equal(d["vid_emb"], emb["vid_emb"])


def test_weight_gradient_workgroup_table(cva):
    """The host-built workgroup -> (problem, tile) table of the batched weight-gradient launch (gemm.hip: tn_xcd_map): every tile
    of every problem exactly once; all tiles that read the same token rows (one problem, one split, one batch group; cut into
    pieces of <= 16) on ONE XCD (block b runs on XCD b % 8); the XCDs evenly loaded."""
    import ctypes as C
    lib = cva.lib.load()
    # the video-side local backward of the ActivityNet workload: input FC 16 x 1, pooling FC 2 (2 groups of 2 x 1), pooling FC 1
    # (2 groups of 3 x 1), W2 / W1 / Wo 3 x 1, QKV 3 x 3; five splits each
    gx, gy, groups, splits = [16, 2, 3, 3, 3, 3, 3], [1, 1, 1, 1, 1, 1, 3], [1, 2, 2, 1, 1, 1, 1], [5] * 7
    n = len(gx)
    arr = lambda v: (C.c_int * len(v))(*v)
    out_item, out_local = (C.c_int * 512)(), (C.c_int * 512)()
    grid = lib.coot_debug_tn_xcd_map(n, arr(gx), arr(gy), arr(groups), arr(splits), out_item, out_local, 512)
    tiles = sum(gx[i] * gy[i] * groups[i] * splits[i] for i in range(n))
    assert grid > 0 and grid % 8 == 0 and tiles <= grid <= 288, (grid, tiles)
    seen, xcd_of, load = set(), {}, [0] * 8
    for b in range(grid):
        it, lo = out_item[b], out_local[b]
        if it < 0:
            continue
        assert 0 <= it < n and 0 <= lo < gx[it] * gy[it] * groups[it] * splits[it]
        assert (it, lo) not in seen
        seen.add((it, lo))
        load[b % 8] += 1
        per = gx[it] * gy[it]
        key = (it, lo // per, (lo % per) // 16)     # problem, (split, group), piece of 16 tiles
        assert xcd_of.setdefault(key, b % 8) == b % 8, key
    assert len(seen) == tiles
    assert max(load) - min(load) <= 4 and max(load) <= 32, load
    # a launch that does not fit the table falls back to the linear order (grid 0)
    assert lib.coot_debug_tn_xcd_map(2, arr([64, 64]), arr([3, 3]), arr([1, 1]), arr([8, 8]), out_item, out_local, 512) == 0


def test_written_gradient_matrices_cover_every_weight_matrix(cva):
    """coot_train_step does not zero the weight-matrix gradients: its backward WRITES them (coot_net_grads_overwrite) and only the
    rest of the arena is zeroed (coot_nets_zero_grads, skip_matrices = 1).  The list of skipped ranges must therefore be exactly
    the parameters a weight-gradient GEMM produces — every *.weight and the pooling weights — and nothing else: a bias or
    LayerNorm vector in it would keep last step's gradient, a matrix missing from it would be accumulated onto garbage."""
    import ctypes as C
    lib = cva.lib.load()
    for oc in H.full_cfgs(2048, 1536, 384, 8, 384, 768) + H.full_cfgs(64, 48, 64, 4, 64, 128):
        tc = cva.TransformerConfig(H.ocfg_to_dict(oc), oc.input_dim)
        cfg = tc.to_c()
        total, table = cva.lib.param_table(cfg)
        offs, sizes = (C.c_int64 * 32)(), (C.c_int64 * 32)()
        n = lib.coot_debug_written_matrices(C.byref(cfg), offs, sizes, 32)
        assert n > 0
        written = sorted((int(offs[i]), int(offs[i]) + int(sizes[i])) for i in range(n))
        assert all(a[1] <= b[0] for a, b in zip(written, written[1:])) and written[-1][1] <= total
        is_matrix = lambda name: name.endswith(".weight") or name.endswith("genpool_w1_head") or name.endswith("genpool_w2_head")
        assert all(is_matrix(name) or name.endswith((".bias", ".gain", "genpool_b1_head", "genpool_b2_head")) for name, _, _ in table)
        matrices = sorted((o, o + int(np.prod(s))) for name, o, s in table if is_matrix(name))
        vectors = [(name, o, int(np.prod(s))) for name, o, s in table if not is_matrix(name)]
        # the per-head pooling weights are one contiguous block per parameter: merge adjacent spans before comparing
        def merge(spans):
            out = []
            for a, b in spans:
                if out and out[-1][1] == a:
                    out[-1] = (out[-1][0], b)
                else:
                    out.append((a, b))
            return out
        assert merge(written) == merge(matrices), (written, matrices)
        for name, o, cnt in vectors:
            assert not any(a < o + cnt and o < b for a, b in written), name


def test_input_stage_layout_and_argument_checks(cva):
    """Host side of the input stages (include/coot_hip.h: COOT_STEP_INPUT_STAGES) and of the block-gather loss: the stage size is the
    normalised features of both sides in whole 128-row tiles + the packed rows' position tables (what csrc/api.hip: layout_saved
    reserves for x^), and bad arguments are refused before anything touches a device."""
    import ctypes as C
    L = cva.lib
    lib = L.load()
    cfgs = H.full_cfgs(2048, 1536, 384, 8, 384, 768)
    sc = L.StepConfig()
    for i, oc in enumerate(cfgs):
        sc.net[i] = cva.TransformerConfig(H.ocfg_to_dict(oc), oc.input_dim).to_c()
    pad = lambda n, a: (n + a - 1) // a * a
    for (B, Nc, Lv, Lc, Lp, Ls) in [(64, 256, 80, 80, 64, 16), (7, 19, 33, 21, 50, 9)]:
        d = L.StepDims(B, Nc, Lv, Lc, Lp, Ls, 5, 5, 0, 0, L.SOURCE_PADDED)
        Tv, Tt = pad(B * Lv + Nc * Lc, 128), pad(B * Lp + Nc * Ls, 128)
        want = pad(Tv * 2048 * 2, 256) + pad(Tt * 1536 * 2, 256) + pad(Tv * 4, 256) + pad(Tt * 4, 256)
        assert lib.coot_step_input_stage_bytes(C.byref(sc), C.byref(d)) == want
    assert lib.coot_step_input_stage_bytes(None, None) == 0
    # two distinct buffers or none
    assert lib.coot_step_set_input_stages(C.c_void_p(4096), C.c_void_p(4096), 1024) != 0 and b"distinct" in lib.coot_last_error()
    assert lib.coot_step_set_input_stages(C.c_void_p(4096), None, 1024) != 0
    assert lib.coot_step_set_input_stages(None, None, 0) == 0
    d = L.StepDims(4, 8, 8, 8, 8, 8, 2, 2, 0, 0, L.SOURCE_PADDED)
    assert lib.coot_step_set_next_batch(None, C.byref(d)) != 0
    assert lib.coot_step_set_next_batch(None, None) == 0
    # the block table of coot_contrastive_fwd_bwd_dp_blocks travels in the kernel arguments: at most DP_MAX_RANKS ranks
    cc = cva.ContrastiveLossConfig(0.2, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0).to_c()
    W = L.DP_MAX_RANKS + 1
    counts, base, ld, down = (C.c_int64 * W)(*([4] * W)), (C.c_int64 * (6 * W))(), (C.c_int64 * 6)(8, 8, 8, 8, 8, 8), (C.c_void_p * 6)()
    dummy = C.c_void_p(4096)
    rc = lib.coot_contrastive_fwd_bwd_dp_blocks(C.byref(cc), W, 0, counts, counts, 64, 32, dummy, base, C.byref(ld), dummy, C.byref(down), dummy, 1 << 20, None)
    assert rc != 0 and b"ranks" in lib.coot_last_error()
    rc = lib.coot_contrastive_fwd_bwd_dp_blocks(C.byref(cc), 2, 2, counts, counts, 64, 32, dummy, base, C.byref(ld), dummy, C.byref(down), dummy, 1 << 20, None)
    assert rc != 0


def test_option_switches_named_in_the_integration_notes_exist(cva):
    """Every A/B / test switch INTEGRATION.md lists is a name coot_set_option knows (host globals: no GPU needed), an unknown name is
    refused with a message, and the read-back ones answer — the documents and the library's switch table do not drift apart."""
    lib = cva.lib.load()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    para = text[text.index("A/B and profiling switches"):]
    para = para[:para.index("`coot_get_option` reads")]
    names = [n for n in re.findall(r"`([a-z0-9_]+)`", para) if not n.startswith("coot_") and n != "gemm_nt"]  # (`gemm_nt`: a kernel named in the text)
    assert {"fused", "cl_small", "cl_col_split", "grad_write", "pack_lazy", "xcd_order", "step_stamps"} <= set(names)
    value = {"cl_col_split": 0, "pack_poison": 0, "grad_poison": 0, "fz_debug": 0, "step_stamps": 0}  # defaults that are not 1
    sizes = {"fused_min_rows", "tn_target_wgs"}  # a size, not a switch: left alone (no read-back)
    for n in names:
        if n in sizes:
            continue
        v = ctypes.c_int32(-1)
        had = lib.coot_get_option(n.encode(), ctypes.byref(v)) == 0
        assert lib.coot_set_option(n.encode(), v.value if had else value.get(n, 1)) == 0, n  # (set to what it is / to its default)
    assert lib.coot_set_option(b"half_tiles", 1) != 0  # removed in round 4
    assert b"unknown option" in lib.coot_last_error()
    v = ctypes.c_int32(-1)
    for n in ("tn_dma", "xcd_order", "tn_mode", "stage_hits", "operand_f16", "fused_attn_launches"):
        assert lib.coot_get_option(n.encode(), ctypes.byref(v)) == 0 and v.value >= 0, n


def test_committed_traffic_profile_was_taken_on_these_kernel_sources():
    """bench.py quotes roofline.traffic from the newest profiles/r*_traffic.json only if that profile carries the hash of the kernel
    sources (csrc/*.hip, csrc/*.h, include/*.h) of the run (tools/sources_hash.py; VERDICT round 5: the line must be able to notice a
    traffic regression).  This test keeps the two from drifting apart silently: after a change to the kernel sources the PMC passes
    (tools/profile_round.sh) are re-run and the new summary committed — or this fails, as the bench line's `traffic: null` +
    `traffic_stale` would tell the driver."""
    import glob
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sources_hash import sources_sha16
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    assert cands
    newest = json.load(open(cands[-1]))
    assert newest.get("kernel_sources_sha16") == sources_sha16(ROOT), (
        f"{os.path.basename(cands[-1])} was collected on kernel sources {newest.get('kernel_sources_sha16')}, the tree is {sources_sha16(ROOT)}: "
        "re-run tools/profile_round.sh on a GPU box and commit profiles/<tag>_traffic.json")
    assert newest["families"]["fused"]["hbm_bytes_per_launch"] > 0
