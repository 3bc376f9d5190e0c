#!/usr/bin/env python3
"""Diagnostic: per-step s_memtime stamps of the panel GEMM (needs a -DPANEL_STAMP build in OCM_VIT_LIB)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vit_ocm_wmsegmentation_amd import _lib

lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
M, N, K = 12608, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, 384
epi = int(sys.argv[2]) if len(sys.argv) > 2 else 100
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
b = torch.randn(N, device=dev)
o = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (0, 1, 100) else torch.bfloat16)
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    assert lib.ocm_op_linear(p(a), p(w), p(b), p(o), p(o), M, N, K, epi, None) == 0
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
assert raw.ocm_debug_panel_stamps(buf, 512) == 0
st = list(buf)
n = max(i for i, v in enumerate(st) if v) + 1
d = [st[i + 1] - st[i] for i in range(n - 1)]
print("stamps:", n, "prologue(A frags + 2 tiles):", d[0])
print("deltas:", d[1:60])
