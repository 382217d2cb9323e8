#!/usr/bin/env python
"""The data-parallel contrastive loss (coot_contrastive_fwd_bwd_dp: every rank scores ITS rows against the gathered global batch) at
the shapes of R ranks of the ActivityNet workload, on one GPU: time per call for R = 1, 2, 4, 8.  python tools/dp_loss_probe.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva

lib, L = cva.lib.load(), cva.lib
D, B, Nc = 384, 64, 256
cfg = cva.ContrastiveLossConfig(0.2, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0).to_c()
sp = torch.cuda.current_stream().cuda_stream
if os.environ.get("CL_COL_SPLIT"):   # forced column splits per strip (0 / unset: by batch size)
    lib.coot_set_option(b"cl_col_split", int(os.environ["CL_COL_SPLIT"]))
RS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]   # python tools/dp_loss_probe.py 8   (one shape, e.g. under rocprofv3 --kernel-trace --stats)
for R in RS:
    nh, nl = R * B, R * Nc
    base_h, base_l = torch.randn(1, 6 * D, device="cuda"), torch.randn(1, 2 * D, device="cuda")
    high = (base_h + 0.8 * torch.randn(nh, 6 * D, device="cuda")).contiguous()   # [n, vid 2D | par 2D | vid_ctx D | par_ctx D]
    low = (base_l + 0.8 * torch.randn(nl, 2 * D, device="cuda")).contiguous()     # [n, clip D | sent D]
    hp, lp, e4 = high.data_ptr(), low.data_ptr(), 4
    sets = (C.c_void_p * 6)(hp, hp + 2 * D * e4, lp, lp + D * e4, hp + 4 * D * e4, hp + 5 * D * e4)
    lds = (C.c_int64 * 6)(6 * D, 6 * D, 2 * D, 2 * D, 6 * D, 6 * D)
    grads = [torch.zeros(B, 2 * D, device="cuda"), torch.zeros(B, 2 * D, device="cuda"), torch.zeros(Nc, D, device="cuda"),
             torch.zeros(Nc, D, device="cuda"), torch.zeros(B, D, device="cuda"), torch.zeros(B, D, device="cuda")]
    down = (C.c_void_p * 6)(*[g.data_ptr() for g in grads])
    loss = torch.zeros(1, device="cuda")
    scratch = torch.empty(lib.coot_contrastive_scratch_bytes(nh, nl, 2 * D, D), dtype=torch.uint8, device="cuda")
    r = R - 1  # the last rank's window
    run = lambda: L.check(lib.coot_contrastive_fwd_bwd_dp(C.byref(cfg), nh, nl, 2 * D, D, C.byref(sets), C.byref(lds), loss.data_ptr(), C.byref(down),
                                                          r * B, B, r * Nc, Nc, scratch.data_ptr(), scratch.numel(), sp), "contrastive_dp")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"R = {R}: global batch {nh} videos / {nl} clips, own rows {B} / {Nc}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (three launches)")
