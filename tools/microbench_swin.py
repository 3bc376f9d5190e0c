#!/usr/bin/env python3
"""Stand-alone timing of the Swin window-attention kernel at the four Swin-T stage shapes (batch 256).
Development tool, GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vit_ocm_wmsegmentation_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for H, heads in ((56, 3), (28, 6), (14, 12), (7, 24)):
    Cc = heads * 32
    T = B * H * H
    qkv = (torch.randn(T, 3 * Cc, device=dev) * 0.5).to(torch.bfloat16)
    ctx = torch.empty(T, Cc, dtype=torch.bfloat16, device=dev)
    table = torch.randn(169, heads, device=dev)
    scratch = torch.empty(heads * (4096 + 7 ** 4), dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shift in (0, 3 if H > 7 else 0):
        def run():
            _lib.check(lib.ocm_op_swin_window_attention(0, C.c_void_p(qkv.data_ptr()), 3 * Cc, C.c_void_p(ctx.data_ptr()), Cc,
                                                        C.c_void_p(table.data_ptr()), C.c_void_p(scratch.data_ptr()), B, H, H,
                                                        7, shift, heads, st))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        mb = (T * 3 * Cc * 2 + T * Cc * 2) / 1e6
        print(f"grid {H}x{H} heads {heads} shift {shift}: {us:8.1f} us  ({mb:.0f} MB -> {mb / us:.2f} TB/s)", flush=True)
