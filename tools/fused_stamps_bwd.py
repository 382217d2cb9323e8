#!/usr/bin/env python
"""Phase timeline of the fused backward chain (block 0): s_memtime stamps at the phase boundaries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cva.lib.check(lib.coot_set_option(b"fused_min_rows", 1))
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
net = H.make_hip_net(cfg, O.make_params(cfg, 3), dropout=0.025)
ts = torch.zeros(64, dtype=torch.int64, device="cuda")
names = ["start", "pool h0", "pool h1", "dz gemm+epi", "ln2 bwd", "dh1", "dz1", "ln1 bwd", "dctx"]
for train in (False, True):
    net.train(train)
    for N in (320, 1):
        x = torch.randn(N, 80, 2048, device="cuda")
        lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
        for _ in range(2):
            pooled, _ = net(x, mask, lens, None, seed=1)
            pooled.sum().backward()
        pooled, _ = net(x, mask, lens, None, seed=1)
        cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
        pooled.sum().backward()
        torch.cuda.synchronize()
        cva.lib.check(lib.coot_debug_timestamps(None))
        t = ts.cpu().numpy()[:len(names)]
        d = (t[1:] - t[:-1])
        print(f"train={train} N={N}: total {(t[-1]-t[0])} ticks; " + ", ".join(f"{n} {int(v)}" for n, v in zip(names[1:], d)))
