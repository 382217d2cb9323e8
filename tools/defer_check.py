"""Diagnostic (round 3): the first RetrievalTrainer of a process against later ones on small networks (d_model 64) — gradient write mode
under NaN poison, and run-to-run reproducibility of six native steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H
lib = cva.lib.load()
dims = (64, 48, 64, 4, 64, 128)
cfgs = H.full_cfgs(*dims)
Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
batch = cva.synthetic.make_batch(7, 6, [1, 2, 3, 4, 2, 1], 12, 10, 9, 6, dims[0], dims[1], ragged=False)
res = []
for k, (poison, write) in enumerate(((1, 1), (0, 1), (0, 0), (0, 1), (0, 0), (0, 1))):
    lib.coot_set_option(b"grad_poison", poison); lib.coot_set_option(b"grad_write", write)
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
    if len(sys.argv) > 1:
        cfg_x.optimizer.adam_eps = float(sys.argv[1])   # e.g. 1e-3: no sign amplification of near-zero gradients
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg_x, mgr)
    g1 = None
    for it in range(6):
        out = tr.train_step_native(batch, seed=100 + it)
        if it == 0:
            torch.cuda.synchronize()
            g1 = torch.cat([n._grad_flat.detach().reshape(-1) for n in mgr.model_dict.values()]).clone()
    torch.cuda.synchronize()
    p = torch.cat([n._flat.detach().reshape(-1) for n in mgr.model_dict.values()])
    res.append((poison, write, [float(v) for v in out], p, g1))
    print(f"run {k}: poison {poison} write {write}: loss {float(out[0]):.6f}; NaN in step-1 gradients: {int(torch.isnan(g1).sum())}, in parameters: {int(torch.isnan(p).sum())}")
for i in range(len(res)):
    for j in range(i + 1, len(res)):
        d = (res[i][3] - res[j][3]).abs(); dg = (res[i][4] - res[j][4]).abs()
        print(f"run {i} vs {j}: params differing by > 1e-6: {float((d > 1e-6).float().mean()):.4f} (max {float(d.max()):.2e}); step-1 gradients max diff {float(dg.max()):.2e} of {float(res[i][4].abs().max()):.2e}")
lib.coot_set_option(b"grad_poison", 0); lib.coot_set_option(b"grad_write", 1)
