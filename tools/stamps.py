#!/usr/bin/env python3
"""Where the cycles of one GEMM workgroup go (development tool, GPU box only). Needs the instrumented build:

    make -C vit-ocm-wmsegmentation_amd/csrc stamps      # -> exp_libs/stamps.so (-DOCM_GEMM_STAMPS)
    python tools/stamps.py

Thread 0 of every workgroup stamps the shader clock at the phase boundaries of gemm_kernel; medians over the
workgroups of the last launch are printed for the ViT-S/16 B=64 shapes. (Counters of different XCDs are not
synchronised: only differences inside a workgroup mean anything.)"""
import ctypes as C, os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["OCM_VIT_LIB"] = os.path.join(ROOT, "exp_libs", "stamps.so")
from vit_ocm_wmsegmentation_amd import _lib
lib = _lib.load(); raw = C.CDLL(os.environ["OCM_VIT_LIB"])
dev = torch.device("cuda:0")
for (M, N, K, epi, name) in [(12608, 1536, 384, 2, "fc1 gelu"), (12608, 1536, 384, 3, "fc1 nogelu"), (12608, 384, 1536, 1, "fc2 resid"), (12608, 384, 384, 1, "proj resid")]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16 if epi >= 2 else torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.ocm_op_linear(0, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()) if epi == 1 else None, C.c_void_p(out.data_ptr()), M, N, K, epi, st)
    torch.cuda.synchronize()
    bm = 128 if (N % 128 == 0 and ((M + 127) // 128) * (N // 128) >= 512) else 64
    tiles = ((M + bm - 1) // bm) * (N // 128); n = min(tiles, 8192)
    buf = np.zeros(n * 8, dtype=np.uint64); raw.ocm_debug_stamps(buf.ctypes.data_as(C.c_void_p), n * 8)
    s = buf.reshape(n, 8).astype(np.int64)
    d = lambda i, j: int(np.median(s[:, j] - s[:, i]))
    print(f"{name:12s} tiles {tiles} ({bm}x128): mainloop(+prologue) {d(0,1)}, stage writes {d(1,2)}, barrier {d(2,3)}, epilogue body {d(3,4)}, store drain {d(4,5)}, lifetime {d(0,5)}", flush=True)
