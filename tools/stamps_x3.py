#!/usr/bin/env python3
"""Phase stamps of the split-bf16 GEMM variants (development tool, GPU box; needs `make -C .../csrc stamps`).
    python tools/stamps_x3.py [variant ...]"""
import ctypes as C, os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["OCM_VIT_LIB"] = os.path.join(ROOT, "exp_libs", "stamps.so")
from vit_ocm_wmsegmentation_amd import _lib
from vit_ocm_wmsegmentation_amd.engine import to_operand
lib = _lib.load(); raw = C.CDLL(os.environ["OCM_VIT_LIB"])
dev = torch.device("cuda:0")
variants = [int(v) for v in sys.argv[1:]] or [0, 4, 1]


TILE = {0: (128, 128), 1: (256, 256), 2: (256, 128), 3: (128, 256), 4: (128, 128), 5: (128, 128), 6: (256, 128),
        7: (64, 128), 8: (64, 128), 9: (64, 128), 10: (64, 64), 11: (128, 192), 12: (128, 192), 14: (128, 192), 15: (128, 128), 16: (128, 192), 17: (128, 192)}
for (M, N, K, epi, name) in [(12608, 1536, 384, 2, "fc1 gelu"), (12608, 384, 1536, 1, "fc2 resid"), (12608, 384, 384, 1, "proj resid")]:
    a = to_operand(torch.randn(M, K, device=dev), 2); w = to_operand(torch.randn(N, K, device=dev) * 0.05, 2)
    b = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for v in variants:
        lib.ocm_debug_knob(0, v)
        for _ in range(3):
            rc = lib.ocm_op_linear(2, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()) if epi == 1 else None, C.c_void_p(out.data_ptr()), M, N, K, epi, st)
            assert rc == 0, lib.ocm_last_error()
        torch.cuda.synchronize()
        bm, bn = TILE[v] if not (v == 0 and N == 384) else (64, 128)  # default dispatch: N = 384 -> 64 x 128 tiles
        if N % bn: continue
        tiles = ((M + bm - 1) // bm) * (N // bn); n = min(tiles, 8192)
        buf = np.zeros(n * 8, dtype=np.uint64); raw.ocm_debug_stamps_linear(buf.ctypes.data_as(C.c_void_p), n * 8)
        s = buf.reshape(n, 8).astype(np.int64)
        d = lambda i, j: int(np.median(s[:, j] - s[:, i]))
        first = f"first tile landed {d(0,6)}, " if v else ""
        if v:
            steps = K // 32
            wv = int(np.median(s[:, 7] & 0xFFFFFFFF)); wb = int(np.median(s[:, 7] >> 32))
            first += f"{steps} steps {d(6,1)} ({d(6,1) // steps}/step; of which own-DMA wait {wv // (steps - 1)}/step, barrier {wb // (steps - 1)}/step), "
        print(f"{name:12s} v{v} tiles {tiles} ({bm}x{bn}): {first}mainloop(+prologue) {d(0,1)}, acc staging {d(1,2)}, barrier {d(2,3)}, epilogue body {d(3,4)}, store drain {d(4,5)}, lifetime {d(0,5)}", flush=True)
lib.ocm_debug_knob(0, 0)

# the fused GEMM + residual + LayerNorm kernels (64 x D tiles, register-staged loop): proj -> norm2, fc2 -> norm1
for (M, D, K, name) in [(12608, 384, 1536, "fc2+LN"), (12608, 384, 384, "proj+LN")]:
    a = to_operand(torch.randn(M, K, device=dev), 2); w = to_operand(torch.randn(D, K, device=dev) * 0.05, 2)
    b = torch.zeros(D, device=dev); g = torch.ones(D, device=dev)
    x = torch.zeros(M, D, dtype=torch.float32, device=dev); xn = torch.zeros(M, D, dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        rc = lib.ocm_op_linear_resid_ln(2, P(a), P(w), P(b), P(x), P(x), P(g), P(b), P(xn), M, D, K, 1e-6, st)
        assert rc == 0, lib.ocm_last_error()
    torch.cuda.synchronize()
    tiles = (M + 63) // 64
    buf = np.zeros(tiles * 8, dtype=np.uint64); raw.ocm_debug_stamps(buf.ctypes.data_as(C.c_void_p), tiles * 8)
    s = buf.reshape(tiles, 8).astype(np.int64)
    d = lambda i, j: int(np.median(s[:, j] - s[:, i]))
    steps = K // 32
    print(f"{name:12s} tiles {tiles} (64x{D}): first tile in LDS {d(0,6)}, {steps} K steps {d(6,1)} ({d(6,1) // steps} per step), "
          f"acc staging {d(1,2)}, barrier {d(2,3)}, epilogue body {d(3,4)}, store drain {d(4,5)}, lifetime {d(0,5)}", flush=True)
