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
    assert lib.coot_version() >= 1


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
