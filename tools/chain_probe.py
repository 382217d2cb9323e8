#!/usr/bin/env python
"""Launch duration (HIP events) of the forward token-tile chain launches (infc_qkv_fwd + post_attn_fwd) of ONE local network call on
its own — no second stream — for 200 / 125 tiles.  Round 3 used it for three experiments that were measured and NOT adopted
(profiles/README.md): an accumulator-layout FF1 epilogue without the fp32 staging, staggered workgroup starts, streaming stores for
the tensors only the backward pass reads.    python tools/chain_probe.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
net = H.make_hip_net(cfg, O.make_params(cfg, 3), dropout=0.025)
net.train(True)
import numpy as np
side = torch.cuda.Stream()
for N in (320, 200, 100):
    x = torch.randn(N, 80, 2048, device="cuda")
    lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
    mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
    for dbg in (0,):
        with torch.no_grad():
            for _ in range(3):
                net(x, mask, lens, None, seed=1)
            torch.cuda.synchronize()
            # shader clock during the launches (round 5): one wave on a side stream samples s_memtime against the 100 MHz counter every 20 us
            nsamp = 400
            mon = torch.zeros(2 * nsamp, dtype=torch.int64, device="cuda")
            with torch.cuda.stream(side):
                cva.lib.check(lib.coot_debug_clock_monitor(mon.data_ptr(), nsamp, 2000, side.cuda_stream), "clock_monitor")
            lib.coot_timing_enable(1)
            for _ in range(10):
                net(x, mask, lens, None, seed=1)
            torch.cuda.synchronize()
            m = mon.cpu().numpy().reshape(-1, 2).astype(np.float64)
            ghz = np.diff(m[:, 1]) / (np.diff(m[:, 0]) * 10.0)
            busy_ghz = ghz[5:120]
            ms, fl, n = C.c_double(), C.c_double(), C.c_int()
            cva.lib.check(lib.coot_timing_collect(5, C.byref(ms), C.byref(fl), C.byref(n)), "timing_collect")
            lib.coot_timing_enable(0)
        print(f"N={N} sequences x 80 frames ({N * 80 // 128} tiles): fused chain launches (infc_qkv_fwd + post_attn_fwd) "
              f"{1e3 * ms.value / 10:.1f} us per forward, {n.value // 10} launches; shader clock while they run {busy_ghz.mean():.2f} GHz "
              f"(p10 {np.percentile(busy_ghz, 10):.2f}, p90 {np.percentile(busy_ghz, 90):.2f}; idle tail {ghz[-40:].mean():.2f})")
