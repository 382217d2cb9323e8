#!/usr/bin/env python
"""Microbenchmark of the two GEMM kernels on the shapes of the ActivityNet workload (runs on the GPU box).
    python tools/gemm_bench.py [--iters 20]
Prints one line per shape: time per launch (HIP events on the launch stream) and algorithmic TFLOP/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import coot_videotext_amd as cva  # noqa: E402

NT_SHAPES = [  # (M, N, K, label)
    (25600, 384, 2048, "in_fc video"), (8192, 384, 1536, "in_fc text"), (25600, 1152, 384, "qkv video"),
    (25600, 384, 384, "wo/ff video"), (25600, 768, 384, "pool1 video"), (25600, 384, 768, "dpool1 video"),
    (25600, 384, 1152, "dz(qkv) video"), (8192, 1152, 384, "qkv text"), (8192, 384, 384, "wo/ff text"),
    (256, 384, 384, "global net"), (256, 1152, 384, "global qkv"),
]
TN_SHAPES = [  # (T, Mo, No, label)
    (25600, 384, 2048, "dW in_fc video"), (8192, 384, 1536, "dW in_fc text"), (25600, 1152, 384, "dW qkv video"),
    (25600, 384, 384, "dW wo/ff video"), (25600, 384, 768, "dW pool1 video"), (8192, 384, 384, "dW wo/ff text"),
    (256, 384, 384, "dW global"),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib = cva.lib.load()
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'kernel':8s} {'shape':28s} {'label':18s} {'us':>9s} {'TF/s':>8s}")
    for M, N, K, label in NT_SHAPES:
        X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        bias = torch.randn(N, device="cuda")
        us = timeit(lambda: lib.coot_gemm_nt(X.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), 0, None, 0,
                                             out.data_ptr(), N, 0, st), args.iters)
        print(f"{'gemm_nt':8s} {str((M, N, K)):28s} {label:18s} {us:9.1f} {2.0 * M * N * K / us / 1e6:8.1f}")
    for T, Mo, No, label in TN_SHAPES:
        A = torch.randn(T, Mo, device="cuda").to(torch.bfloat16)
        B = torch.randn(T, No, device="cuda").to(torch.bfloat16)
        Cm = torch.zeros(Mo, No, device="cuda")
        ws = torch.empty(lib.coot_gemm_tn_workspace_bytes(T, Mo, No), dtype=torch.uint8, device="cuda")
        us = timeit(lambda: lib.coot_gemm_tn(A.data_ptr(), Mo, B.data_ptr(), No, T, Mo, No, Cm.data_ptr(), No, ws.data_ptr(), ws.numel(), st), args.iters)
        print(f"{'gemm_tn':8s} {str((T, Mo, No)):28s} {label:18s} {us:9.1f} {2.0 * T * Mo * No / us / 1e6:8.1f}")


if __name__ == "__main__":
    main()
