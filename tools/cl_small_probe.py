#!/usr/bin/env python
"""The (vid, par) contrastive terms of the train step (64 rows x 768 features, cluster terms on) called back to back: run under
rocprofv3 --kernel-trace --stats to get the durations of cl_small_kernel / cl_finish_kernel (or cl_norm / cl_half / cl_finish with
CL_SMALL=0).  python tools/cl_small_probe.py [n d]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coot_videotext_amd as cva

lib, L = cva.lib.load(), cva.lib
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 768)
lib.coot_set_option(b"cl_small", int(os.environ.get("CL_SMALL", "1")))
cfg = cva.lib.ContrastiveConfig(0.2, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
base = torch.randn(1, d, device="cuda")
a = base + 0.6 * torch.randn(n, d, device="cuda")
ts = [a, a + 0.9 * torch.randn(n, d, device="cuda")] + [torch.zeros(16, 32, device="cuda") for _ in range(4)]
grads = [torch.zeros_like(t) for t in ts]
loss = torch.zeros(1, device="cuda")
scratch = torch.empty(lib.coot_contrastive_scratch_bytes(n, 16, d, 32), dtype=torch.uint8, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
run = lambda: L.check(lib.coot_contrastive_fwd_bwd_part(C.byref(cfg), n, 16, d, 32, *[t.data_ptr() for t in ts], loss.data_ptr(),
                                                        *[g.data_ptr() for g in grads], scratch.data_ptr(), scratch.numel(), 1, sp), "part")
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    run()
e1.record(); torch.cuda.synchronize()
print(f"n={n} d={d} cl_small={os.environ.get('CL_SMALL', '1')}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per call")
