#!/usr/bin/env python
"""Forward of one local network at N sequences x 80 frames, a few times — the workload of tools/chain_probe.py as a plain loop for
rocprofv3 --pmc passes (round 5: which SQ counters of post_attn_fwd / infc_qkv_fwd change between 62 and 200 resident tiles?).
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace -d out -- python tools/chain_pmc_probe.py 320"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
cva.lib.load()
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
net = H.make_hip_net(cfg, O.make_params(cfg, 3), dropout=0.025)
net.train(True)
x = torch.randn(N, 80, 2048, device="cuda")
lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
with torch.no_grad():
    for _ in range(6):
        net(x, mask, lens, None, seed=1)
torch.cuda.synchronize()
