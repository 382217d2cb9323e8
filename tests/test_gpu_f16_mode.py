"""The IEEE-half build of the library (libcoot_hip_f16.so: csrc/common.h -DCOOT_OPERAND_F16; BASELINE.json configs[3] "fp16 MFMA path",
the arithmetic of the reference's GPU path — fp16 autocast, coot/trainer_retrieval.py:264) against fixtures the unmodified reference
wrote.  One library per process, so each case runs in a child process with COOT_OPERAND=f16:
  * configs[3] as worded (Dv = 2816, 64 videos x 8 clips), TRAIN-mode forward with the library's dropout masks injected into the
    reference (bench_yc2_2d3d_2816_train): every embedding row cosine > 1 - 1e-3, and the same step under another seed fails the bound;
  * the ActivityNet eval fixture through encode_visual / encode_text (the reference's call surface);
  * the build is forward-only: its backward entry points refuse (no GradScaler), the bf16 dtype is refused, and the bf16 build
    refuses dtype f16 — nothing is silently computed in the other format.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["COOT_ROOT"])
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H
from tests.test_gpu_train_parity import _case, _step_forward
golden = os.path.join(os.environ["COOT_ROOT"], "tests", "golden")
out = {"operand": cva.lib.operand()}
lib = cva.lib.load()
# ---- configs[3] as worded: train-mode forward, injected masks ----
name = "bench_yc2_2d3d_2816_train"
g, cfgs, Ps, b = _case(golden, name)
p, seed = float(g["train_p"]), int(g["train_step_seed"])
cfg, mgr = H.make_manager(cfgs, Ps, dropout=p, cc_weight=float(g["cc_weight"]))
mgr.set_all_models_train()
trainer = cva.RetrievalTrainer(cfg, mgr)
batch = cva.synthetic.batch_from_numpy(b)
emb = _step_forward(torch, cva, trainer, batch, seed)
out["cos_2816"] = {k: float(H.cosine_rows(v, g[k]).min()) for k, v in emb.items()}
emb2 = _step_forward(torch, cva, trainer, batch, seed + 1)
out["ctrl_2816"] = float(min(H.cosine_rows(emb2[k], g[k]).min() for k in emb2))
# the backward is refused
try:
    trainer.train_step_native(batch, do_optimizer=False, seed=seed)
    out["bwd_refused"] = False
except RuntimeError as e:
    out["bwd_refused"] = "forward-only" in str(e)
# ---- ActivityNet eval fixture through the reference's call surface ----
g = dict(np.load(os.path.join(golden, "bench_anet.npz")))
seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph)
Ps = [O.make_params(cfgs[i], seed + 10 * i, scale=float(g["param_scale"])) for i in range(4)]
bb = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=bool(int(g["ragged"])), corr=0.5)
cfg, mgr = H.make_manager(cfgs, Ps)
mgr.set_all_models_eval()
batch = cva.synthetic.batch_from_numpy(bb)
with torch.no_grad():
    vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
torch.cuda.synchronize()
got = {"vid_emb": vis.vid_emb, "clip_emb": vis.clip_emb, "vid_context": vis.vid_context, "par_emb": txt.par_emb, "sent_emb": txt.sent_emb, "par_context": txt.par_context}
out["cos_anet"] = {k: float(H.cosine_rows(v.cpu().numpy(), g[k]).min()) for k, v in got.items()}
out["max_abs_anet"] = {k: float(np.abs(v.cpu().numpy() - g[k]).max()) for k, v in got.items()}
# ---- the other 16-bit format is refused ----
net = mgr.model_dict["net_video_global"]
net.set_compute_dtype("bf16" if out["operand"] == "f16" else "f16")
try:
    mgr.encode_visual(batch)
    out["other_format_refused"] = False
except RuntimeError as e:
    out["other_format_refused"] = "operands" in str(e)
print("RESULT " + json.dumps(out))
'''


def _child(operand):
    env = dict(os.environ, COOT_OPERAND=operand, COOT_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(line[-1][7:])


def test_f16_operand_build_forward_matches_the_reference():
    out = _child("f16")
    print(json.dumps(out, indent=1))
    assert out["operand"] == "f16"
    for k, c in out["cos_2816"].items():
        assert c > 1 - 1e-3, (k, c)
    assert out["ctrl_2816"] < 0.995, out["ctrl_2816"]   # another seed's masks must fail the bound
    for k, c in out["cos_anet"].items():
        assert c > 1 - 1e-3, (k, c)
    assert out["bwd_refused"] and out["other_format_refused"]


def test_bf16_build_refuses_the_f16_dtype():
    import torch
    import coot_videotext_amd as cva
    from oracle import coot_oracle as O
    from tests import helpers as H
    if cva.lib.operand() != "bf16":
        pytest.skip("this process loaded the f16 build")
    dims = (64, 48, 64, 4, 64, 128)
    cfgs = H.full_cfgs(*dims)
    Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
    batch = cva.synthetic.make_batch(7, 6, [1, 2, 3, 4, 2, 1], 12, 10, 9, 6, dims[0], dims[1], ragged=False)
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.0)
    mgr.set_all_models_eval()
    mgr.model_dict["net_video_local"].set_compute_dtype("f16")
    with pytest.raises(RuntimeError, match="operands"):
        mgr.encode_visual(batch)
