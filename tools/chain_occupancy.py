#!/usr/bin/env python
"""Forward chain launches (infc_qkv_fwd + post_attn_fwd) of one local-network call on their own at 200 / 125 / 62 / 8 tiles: how a
tile's latency grows with the number of resident tiles, for whatever build COOT_HIP_LIB names (tools/build_fused_variant.sh)."""
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
ZERO = os.environ.get("COOT_PROBE_ZERO") == "1"  # all-zero inputs and weights: same instruction stream, minimal datapath toggling
P = O.make_params(cfg, 3)
if ZERO:
    import numpy as np
    P = {k: np.zeros_like(v) for k, v in P.items()}
net = H.make_hip_net(cfg, P, dropout=0.025)
net.train(True)
out = []
for N in (320, 200, 100, 13):
    x = torch.zeros(N, 80, 2048, device="cuda") if ZERO else torch.randn(N, 80, 2048, device="cuda")
    lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
    mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            net(x, mask, lens, None, seed=1)
        torch.cuda.synchronize()
        lib.coot_timing_enable(1)
        for _ in range(10):
            net(x, mask, lens, None, seed=1)
        torch.cuda.synchronize()
        ms, fl, n = C.c_double(), C.c_double(), C.c_int()
        cva.lib.check(lib.coot_timing_collect(5, C.byref(ms), C.byref(fl), C.byref(n)), "timing_collect")
        lib.coot_timing_enable(0)
    out.append(f"{(N * 80 + 127) // 128} tiles {1e3 * ms.value / 10:.1f} us")
print(os.path.basename(os.environ.get("COOT_HIP_LIB", "default")) + (" zero-data" if ZERO else ""), "|", " | ".join(out))
