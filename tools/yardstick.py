#!/usr/bin/env python3
"""Yardstick (development only): what the vendor GEMM library reaches on the ViT-S/16 B=64 projection
shapes, timed with torch.matmul (hipBLASLt / rocBLAS underneath). Not part of the product path."""
import torch

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, M, N, K in [("qkv", 12608, 1152, 384), ("proj", 12608, 384, 384), ("fc1", 12608, 1536, 384),
                      ("fc2", 12608, 384, 1536), ("fc1_B128", 25216, 1536, 384), ("vitb_fc1", 73856, 3072, 768)]:
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: torch.mm(a, w.t(), out=out))
    t2 = timeit(lambda: torch.nn.functional.linear(a, w, bias))
    print(f"{name:10s} M={M} N={N} K={K}: mm {t:7.2f} us = {2.0*M*N*K/t/1e6:6.1f} TFLOP/s; linear+bias {t2:7.2f} us", flush=True)
