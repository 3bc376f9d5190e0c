#!/usr/bin/env python3
"""Development tool (GPU box; `make -C vit-ocm-wmsegmentation_amd/csrc stamps`): phase stamps of the LDS-DMA nn.Linear kernels at the
row counts of a one-tile-per-call forward (M = 197): cycles until the first tile lands, per K step (own-DMA wait, barrier),
epilogue. DESIGN.md section 3.11 quotes its output for the four-wave tile (850 cycles per K step for 384 cycles of MFMA).
    python tools/stamps_b1.py"""
import ctypes as C, os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["OCM_VIT_LIB"] = os.path.join(ROOT, "exp_libs", "stamps.so")
from vit_ocm_wmsegmentation_amd import _lib
from vit_ocm_wmsegmentation_amd.engine import to_operand
lib = _lib.load(); raw = C.CDLL(os.environ["OCM_VIT_LIB"])
dev = torch.device("cuda:0")
for (M, N, K, epi, name, bm, bn) in [(197, 1536, 384, 2, "fc1 gelu", 64, 128), (197, 384, 384, 1, "proj resid", 64, 128)]:
    a = to_operand(torch.randn(M, K, device=dev), 2); w = to_operand(torch.randn(N, K, device=dev) * 0.05, 2)
    b = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = lib.ocm_op_linear(2, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()) if epi == 1 else None, C.c_void_p(out.data_ptr()), M, N, K, epi, st)
        assert rc == 0, lib.ocm_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.ocm_op_linear(2, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()) if epi == 1 else None, C.c_void_p(out.data_ptr()), M, N, K, epi, st)
    e1.record(); torch.cuda.synchronize()
    tiles = ((M + bm - 1) // bm) * (N // bn)
    buf = np.zeros(tiles * 8, dtype=np.uint64); raw.ocm_debug_stamps_linear(buf.ctypes.data_as(C.c_void_p), tiles * 8)
    s = buf.reshape(tiles, 8).astype(np.int64)
    d = lambda i, j: int(np.median(s[:, j] - s[:, i]))
    steps = K // 32
    wv = int(np.median(s[:, 7] & 0xFFFFFFFF)); wb = int(np.median(s[:, 7] >> 32))
    print(f"{name:12s} M={M} tiles {tiles} ({bm}x{bn}) {e0.elapsed_time(e1)/20*1e3:.1f} us/launch back to back: first tile landed {d(0,6)}, {steps} steps {d(6,1)} ({d(6,1)//steps}/step; own-DMA wait {wv//(steps-1)}/step, barrier {wb//(steps-1)}/step), acc staging {d(1,2)}, barrier {d(2,3)}, epilogue body {d(3,4)}, store drain {d(4,5)}, lifetime {d(0,5)} cycles")
