"""The fp32 REFERENCE MODE of the library (coot_net_config.dtype = COOT_DTYPE_F32) against fixtures the unmodified
reference wrote: with every activation, weight and accumulation in fp32 the embeddings must agree with the reference's to fp32
round-off — where the bf16 fast path agrees to ~1e-3 of the output scale.  Turns "is this difference bf16 noise or a logic error?"
into a measurement: a wrong mask, position, residual or pooling rule shows up at 1e-2 ... 1 in BOTH modes, bf16 rounding only in one.
"""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

# eval-mode fixtures (oracle/gen_golden.py): a 6-video batch at the paper dims, the benchmark shapes (fixed, ragged, Cmax = 64, 2-layer local networks)
CASES = ["full_anet", "bench_anet", "bench_anet_ragged", "bench_hbm_stress", "bench_yc2_100m_2layer"]
KEYS = ("vid_emb", "clip_emb", "vid_context", "par_emb", "sent_emb", "par_context")


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _load(golden_dir, name):
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
    cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph, layers=int(g["layers"]) if "layers" in g else 1)
    Ps = [O.make_params(cfgs[i], seed + 10 * i, scale=float(g["param_scale"])) for i in range(4)]
    b = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=bool(int(g["ragged"])), corr=0.5)
    return g, cfgs, Ps, b


def _embeddings(torch, mgr, batch):
    with torch.no_grad():
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
    torch.cuda.synchronize()
    return {"vid_emb": vis.vid_emb, "clip_emb": vis.clip_emb, "vid_context": vis.vid_context, "par_emb": txt.par_emb, "sent_emb": txt.sent_emb,
            "par_context": txt.par_context}


@pytest.mark.parametrize("name", CASES)
def test_f32_reference_mode_matches_the_reference_to_round_off(env, golden_dir, name):
    torch, cva = env
    g, cfgs, Ps, b = _load(golden_dir, name)
    cfg, mgr = H.make_manager(cfgs, Ps, cc_weight=float(g["cc_weight"]) if "cc_weight" in g else 0.01)
    mgr.set_all_models_eval()
    batch = cva.synthetic.batch_from_numpy(b)
    fast = {k: v.cpu().numpy().astype(np.float64) for k, v in _embeddings(torch, mgr, batch).items()}
    for net in mgr.model_dict.values():
        net.set_compute_dtype("f32")
    ref = {k: v.cpu().numpy().astype(np.float64) for k, v in _embeddings(torch, mgr, batch).items()}
    worst_abs, worst_fast = 0.0, 0.0
    for k in KEYS:
        want = g[k].astype(np.float64)
        scale = max(1.0, float(np.abs(want).max()))
        e32, e16 = float(np.abs(ref[k] - want).max()), float(np.abs(fast[k] - want).max())
        worst_abs, worst_fast = max(worst_abs, e32 / scale), max(worst_fast, e16 / scale)
        print(f"[{name}] {k}: max |f32 mode - reference| = {e32:.2e}, max |bf16 path - reference| = {e16:.2e} (|reference| max {np.abs(want).max():.3f})")
        assert e32 <= 1e-5 * scale, (k, e32)
    # the two modes differ by the bf16 rounding of the fast path and by nothing else
    assert worst_fast > 10 * worst_abs, (worst_fast, worst_abs)
    # a backward pass in this mode must refuse, not silently run the bf16 kernels
    net = mgr.model_dict["net_video_global"]
    x = torch.randn(2, 3, cfgs[1].input_dim, device="cuda", requires_grad=True)
    out, _ = net(x, None, torch.tensor([3, 2], device="cuda"), torch.randn(2, cfgs[1].hidden_dim, device="cuda"))
    with pytest.raises(RuntimeError, match="forward-only"):
        out.sum().backward()
    net.train()
    with pytest.raises(RuntimeError, match="eval-mode"):
        net(x.detach(), None, torch.tensor([3, 2], device="cuda"), torch.randn(2, cfgs[1].hidden_dim, device="cuda"))
