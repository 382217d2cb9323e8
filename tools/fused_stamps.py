#!/usr/bin/env python
"""Phase timeline of the fused post-attention kernel (block 0): s_memtime stamps at the phase boundaries."""
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
names = ["start", "load ctx", "gemm wo", "epi wo", "ln1", "gemm w1", "epi w1", "gemm+epi w2", "ln2", "pool h0", "pool h1"]
for train in (False, True):
    net.train(train)
    for N in (320, 1):
        x = torch.randn(N, 80, 2048, device="cuda")
        lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
        for dbg in (0, 7):
            cva.lib.check(lib.coot_set_option(b"fz_debug", dbg))
            with torch.no_grad():
                for _ in range(3):
                    net(x, mask, lens, None, seed=1)
                cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
                net(x, mask, lens, None, seed=1)
                torch.cuda.synchronize()
                cva.lib.check(lib.coot_debug_timestamps(None))
            t = ts.cpu().numpy()[:len(names)]
            d = (t[1:] - t[:-1])
            print(f"train={train} N={N} dbg={dbg}: total {(t[-1]-t[0])} ticks; " + ", ".join(f"{n} {int(v)}" for n, v in zip(names[1:], d)))
            ti = ts.cpu().numpy()[48:54]
            if ti[0] and dbg == 0:
                di = ti[1:] - ti[:-1]
                print(f"    input FC + QKV kernel: total {int(ti[-1] - ti[0])}; K loop {int(di[0])}, epilogue (h0, GELU, pe, z0) {int(di[1])}, "
                      f"q {int(di[2])}, k {int(di[3])}, v {int(di[4])}")
cva.lib.check(lib.coot_set_option(b"fz_debug", 0))
