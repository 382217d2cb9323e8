#!/usr/bin/env python
"""Ad-hoc probes of gemm_nt launch time vs shape/epilogue (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import coot_videotext_amd as cva
from tools.gemm_bench import timeit

lib = cva.lib.load()
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(25600, 384, 384), (12800, 384, 384), (6400, 384, 384), (3200, 384, 384), (25600, 128, 384), (25600, 384, 64),
                  (25600, 384, 128), (25600, 384, 768), (65536, 384, 384)]:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    out32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    bias = torch.randn(N, device="cuda")
    t_b = timeit(lambda: lib.coot_gemm_nt(X.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0, out.data_ptr(), N, 0, st), 20)
    t_nb = timeit(lambda: lib.coot_gemm_nt(X.data_ptr(), K, W.data_ptr(), K, M, N, K, None, 0, None, 0, out.data_ptr(), N, 0, st), 20)
    t_f32 = timeit(lambda: lib.coot_gemm_nt(X.data_ptr(), K, W.data_ptr(), K, M, N, K, None, 0, None, 0, out32.data_ptr(), N, 1, st), 20)
    t_gelu = timeit(lambda: lib.coot_gemm_nt(X.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), 1, None, 0, out.data_ptr(), N, 0, st), 20)
    print(f"{(M, N, K)}: bias {t_b:.1f} us | no-bias {t_nb:.1f} | f32-out {t_f32:.1f} | bias+gelu {t_gelu:.1f}   ({2.0*M*N*K/t_nb/1e6:.0f} TF/s no-bias)")
# copy bandwidth reference
a = torch.empty(25600 * 384, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
print("torch copy 19.6MB:", timeit(lambda: b.copy_(a), 20), "us")
