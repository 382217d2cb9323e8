#!/usr/bin/env python
"""Short self-attention forward on its own (coot_attn_fwd: one workgroup per (sequence, head)): time per launch at the video-side and
text-side shapes of the ActivityNet workload.  python tools/attn_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva

lib = cva.lib.load()
st = torch.cuda.current_stream().cuda_stream
for name, N, L in (("video side: 320 sequences x 80", 320, 80), ("text side, paragraphs: 64 x 64", 64, 64), ("text side, sentences: 256 x 16", 256, 16)):
    H, dh = 8, 48
    D = H * dh
    qkv = (torch.randn(N * L, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(N * L, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(N * L * H, dtype=torch.float32, device="cuda")
    lens = torch.full((N,), L, dtype=torch.int64, device="cuda")
    run = lambda: cva.lib.check(lib.coot_attn_fwd(qkv.data_ptr(), N, L, H, dh, lens.data_ptr(), out.data_ptr(), lse.data_ptr(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    mb = (qkv.numel() + out.numel()) * 2 / 1e6
    print(f"{name}: {us:.1f} us per launch, {mb:.0f} MB of q | k | v read and output written -> {mb / us:.2f} TB/s" .replace(" TB/s", "e-0 TB/s") if False else f"{name}: {us:.1f} us per launch, {mb:.0f} MB of q | k | v read and output written -> {mb / us / 1e6 * 1e6:.2f} MB/us (= TB/s)")
