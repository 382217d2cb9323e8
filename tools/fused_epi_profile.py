#!/usr/bin/env python
"""Sub-phase cycle counters of the FF1 epilogue of the fused forward chain (wave 0 of block 0).  Needs a library built with
-DFZ_PROFILE_EPI (EXTRA_FLAGS=-DFZ_PROFILE_EPI bash coot-videotext_amd/csrc/build.sh); rebuild without it afterwards."""
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
names = ["barrier 1", "acc -> staging", "barrier 2", "staging -> regs", "3 chunk bodies (+ stores)", "tile write-back + prefetch"]
for train in (False, True):
    net.train(train)
    for N in (320, 1):
        x = torch.randn(N, 80, 2048, device="cuda")
        lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                net(x, mask, lens, None, seed=1)
            cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
            net(x, mask, lens, None, seed=1)
            torch.cuda.synchronize()
            cva.lib.check(lib.coot_debug_timestamps(None))
        t = ts.cpu().numpy()
        print(f"train={train} N={N}: epi w1 total {int(t[6] - t[5])}; " + ", ".join(f"{n} {int(v)}" for n, v in zip(names, t[32:38])))
