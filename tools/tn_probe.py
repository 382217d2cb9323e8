#!/usr/bin/env python
"""Weight-gradient launch of one local-network backward pass on its own (coot_gemm_tn_batch): time per launch, TF/s, and the k-loop
phase times of tile (0, 0) of the first problem.  python tools/tn_probe.py [video|text] [xcd_order] [tn_dma]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva
from coot_videotext_amd.lib import TnProblem

side = sys.argv[1] if len(sys.argv) > 1 else "video"
lib = cva.lib.load()
if len(sys.argv) > 2:
    cva.lib.check(lib.coot_set_option(b"xcd_order", int(sys.argv[2])))
if len(sys.argv) > 3:
    cva.lib.check(lib.coot_set_option(b"tn_dma", int(sys.argv[3])))
if len(sys.argv) > 4:
    cva.lib.check(lib.coot_set_option(b"tn_target_wgs", int(sys.argv[4])))
T, Din = (25600, 2048) if side == "video" else (8192, 1536)
D = 384
dev = "cuda"
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
f32 = lambda *s: torch.zeros(*s, device=dev)
keep = []
def prob(A, lda, B, ldb, Mo, No, groups=1, zA=0, zB=0, overwrite=0):
    Cm = f32(groups, Mo, No); keep.extend([A, B, Cm])
    return TnProblem(A.data_ptr(), lda, B.data_ptr(), ldb, T, Mo, No, Cm.data_ptr(), No, None, overwrite, groups, zA, zB, Mo * No)
probs = [
    prob(bf(T, D), D, bf(T, Din), Din, D, Din, overwrite=1),            # input FC:  dh0^T . xhat   (first: carries the stamps)
    prob(bf(T, 768), 768, bf(T, D), D, 384, 192, groups=2, zA=384, zB=192),   # pooling FC 2 (per head)
    prob(bf(T, D), D, bf(T, 768), 768, 384, 384, groups=2, zA=0, zB=384),     # pooling FC 1
    prob(bf(T, D), D, bf(T, D), D, D, D), prob(bf(T, D), D, bf(T, D), D, D, D), prob(bf(T, D), D, bf(T, D), D, D, D),  # W2, W1, Wo
    prob(bf(T, 3 * D), 3 * D, bf(T, D), D, 3 * D, D),                  # QKV
]
arr = (TnProblem * len(probs))(*probs)
flops = sum(2.0 * T * p.Mo * p.No * p.groups for p in probs)
ws = torch.empty(8 * sum(p.Mo * p.No * p.groups for p in probs), device=dev)
stamps = torch.zeros(64 + 2 * 512, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
run = lambda s=None: cva.lib.check(lib.coot_gemm_tn_batch(arr, len(probs), ws.data_ptr(), ws.numel() * 4, s, st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for _ in range(N):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print(f"{side}: T={T} {len(probs)} problems, {flops / 1e9:.1f} GFLOP, {ms * 1e3:.1f} us per launch (incl. split reduction), {flops / ms / 1e9:.0f} TF/s")
run(stamps.data_ptr()); torch.cuda.synchronize()
s = stamps.cpu().numpy()
n = max(int(s[0]), 1)
print(f"  tile (0,0) of the input-FC problem: {n} k-steps, {s[5] / n:.0f} shader clocks per k-step: phases {s[1] / n:.0f}, {s[2] / n:.0f}, {s[3] / n:.0f}, {s[4] / n:.0f}"
      " (register-staged: load issue, LDS reads + MFMA, load wait + LDS stores, barrier; LDS-DMA: wait for the stage, barrier, DMA issue, LDS reads + MFMA)")
if s[8] or s[9] or s[10]:
    print(f"  loader wave 0 per k-step: wait for the stage to land {s[8] / n:.0f}, barrier {s[9] / n:.0f}, DMA issue {s[10] / n:.0f}")
if s[16:40].any():
    t0 = min(int(v) for v in s[16:40] if v)
    print("  k-step 20, barrier arrival / release per wave (clocks after the first arrival; waves 8-11 load): " +
          ", ".join(f"w{w} {int(s[16 + 2 * w]) - t0}/{int(s[17 + 2 * w]) - t0}" for w in range(12)))
sp = s[64:].reshape(-1, 2)
sp = sp[sp[:, 0] > 0]
if len(sp):
    t0 = sp[:, 0].min()
    st, en = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0   # us
    import numpy as np
    order = np.argsort(en)
    print(f"  {len(sp)} workgroups (MFMA wave 0, us after the first start): start max {st.max():.1f}; end min / median / max {en.min():.1f} / {np.median(en):.1f} / {en.max():.1f}")
    blk = np.nonzero(s[64:].reshape(-1, 2)[:, 0] > 0)[0]
    for x in range(8):
        m = (blk % 8) == x
        if m.any():
            print(f"    XCD {x}: {int(m.sum())} workgroups, end median {np.median(en[m]):.1f}, max {en[m].max():.1f}")
if len(sp) and s[5]:
    b0 = s[64:66]
    print(f"  workgroup 0: {int(s[5])} s_memtime clocks in {(b0[1] - b0[0]) / 100.0:.1f} us of the 100 MHz counter -> {s[5] / ((b0[1] - b0[0]) * 10.0):.2f} clocks per ns")
