#!/usr/bin/env python
"""Profiling aid for the fused chains: runs the ANet local network forward at a few sizes with parts of the fused kernel
switched off (coot_set_option("fz_debug", bits)); read the per-launch durations with rocprofv3 --kernel-trace + tools/rocpd_stats.py
(or the per-call listing this script prints from torch events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cva.lib.check(lib.coot_set_option(b"fused_min_rows", 1))
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
P = O.make_params(cfg, 3)
for train in (False, True):
    net = H.make_hip_net(cfg, P, dropout=0.025)
    net.train(train)
    for N in (320, 1):
        L = 80
        x = torch.randn(N, L, 2048, device="cuda")
        lens = torch.full((N,), L, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, L, dtype=torch.bool, device="cuda")
        for dbg in (0, 1, 2, 3, 4, 7):
            cva.lib.check(lib.coot_set_option(b"fz_debug", dbg))
            with torch.no_grad():
                for _ in range(3):
                    net(x, mask, lens, None, seed=1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    net(x, mask, lens, None, seed=1)
                e1.record()
                torch.cuda.synchronize()
            print(f"train={train} N={N} T={N*L} fz_debug={dbg}: net fwd {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
cva.lib.check(lib.coot_set_option(b"fz_debug", 0))
