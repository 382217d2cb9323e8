#!/usr/bin/env python
"""Noise level of a short training run on ragged packed batches (plain steps twice) next to the same run with the input-stage lookahead
(train_step_native(next_batch = ...)): per-step losses of the three runs.  python tools/lookahead_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

dims = (256, 192, 384, 8, 384, 768)
mk = lambda s: cva.synthetic.make_batch(s, 12, cva.synthetic.anet_like_counts(50 + s, 12), 40, 40, 32, 16, dims[0], dims[1], ragged=True, packed=True)
cfgs = H.full_cfgs(*dims)
Ps = [O.make_params(cfgs[i], 1 + i, scale=0.02) for i in range(4)]
batches = [mk(20 + i) for i in range(5)]
decoy = mk(99)
announce = [batches[1], decoy, batches[3], None, None]
eps = float(os.environ.get("ADAM_EPS", "1e-3"))
for name, look in (("plain", False), ("plain again", False), ("lookahead", True), ("lookahead again", True)):
    cfg_x, mgr = H.make_manager(cfgs, Ps, dropout=0.1, cc_weight=0.01)
    cfg_x.optimizer.adam_eps = eps
    mgr.set_all_models_train()
    tr = cva.RetrievalTrainer(cfg_x, mgr)
    tr.lookahead_min_stage_bytes = 0
    ls = []
    for it, b in enumerate(batches):
        out = tr.train_step_native(b, seed=100 + it, next_batch=announce[it] if look else None)
        ls.append(float(out[0]))
    torch.cuda.synchronize()
    print(f"{name:16s}", " ".join(f"{v:.7f}" for v in ls))
