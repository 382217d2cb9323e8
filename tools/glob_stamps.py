#!/usr/bin/env python
"""Phase timeline of the single-launch global network forward (block 0): s_memtime stamps at the phase boundaries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cfg = O.NetConfig(input_dim=384, hidden_dim=384, num_heads=8, ff_dim=384, use_input_fc=False, use_context=True, pooler="avg_special")
net = H.make_hip_net(cfg, O.make_params(cfg, 3), dropout=0.025)
ts = torch.zeros(64, dtype=torch.int64, device="cuda")
bnames = ["start", "context chain", "one-query attention bwd", "dhidden (q proj dX)", "k, v proj dX + avg pool bwd", "encoder chain", "self-attention bwd",
          "QKV dX", "input LN bwd"]
names = ["start", "LN + pe", "q", "k", "v", "self attention", "encoder chain", "context k, v (+ avg pool)", "query tile + q", "context attention", "context chain"]
for train in (False, True):
    net.train(train)
    for (N, L) in ((64, 4), (64, 8), (64, 27)):
        x = torch.randn(N, L, 384, device="cuda")
        hid = torch.randn(N, 384, device="cuda")
        lens = torch.full((N,), L, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, L, dtype=torch.bool, device="cuda")
        xg = x.clone().requires_grad_(True)
        hg = hid.clone().requires_grad_(True)
        cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
        pooled, _ = net(xg, mask, lens, hg, seed=1)
        pooled.sum().backward()
        torch.cuda.synchronize()
        cva.lib.check(lib.coot_debug_timestamps(None))
        tb = ts.cpu().numpy()[16:16 + len(bnames)]
        print(f"train={train} N={N} L={L} BACKWARD block 0 total {tb[-1] - tb[0]} shader cycles; " + ", ".join(f"{n} {int(v)}" for n, v in zip(bnames[1:], tb[1:] - tb[:-1])))
        with torch.no_grad():
            for _ in range(3):
                net(x, mask, lens, hid, seed=1)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(20):
                net(x, mask, lens, hid, seed=1)
            ev1.record()
            torch.cuda.synchronize()
            cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
            net(x, mask, lens, hid, seed=1)
            torch.cuda.synchronize()
            cva.lib.check(lib.coot_debug_timestamps(None))
        t = ts.cpu().numpy()[:len(names)]
        d = (t[1:] - t[:-1])
        print(f"train={train} N={N} L={L}: {ev0.elapsed_time(ev1) * 50:.1f} us per forward call (python included); block 0 total {(t[-1]-t[0])} shader cycles; "
              + ", ".join(f"{n} {int(v)}" for n, v in zip(names[1:], d)))
