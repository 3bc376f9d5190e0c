#!/usr/bin/env python3
"""Development tool (GPU box; `make -C .../csrc stamps`): per-wave timeline of the middle K step of an LDS-DMA nn.Linear kernel.
    python tools/wstamps.py VARIANT [M N K epi]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["OCM_VIT_LIB"] = os.path.join(ROOT, "exp_libs", "stamps.so")
from vit_ocm_wmsegmentation_amd import _lib
from vit_ocm_wmsegmentation_amd.engine import to_operand
lib = _lib.load(); raw = C.CDLL(os.environ["OCM_VIT_LIB"])
dev = torch.device("cuda:0")
v = int(sys.argv[1])
M, N, K, epi = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else (12608, 384, 1536, 1)
a = to_operand(torch.randn(M, K, device=dev), 2); w = to_operand(torch.randn(N, K, device=dev) * 0.05, 2)
b = torch.zeros(N, device=dev)
out = torch.zeros(M, N, dtype=torch.float32 if epi < 2 else torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.ocm_debug_knob(0, v)
for _ in range(3):
    rc = lib.ocm_op_linear(2, C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()) if epi == 1 else None, C.c_void_p(out.data_ptr()), M, N, K, epi, st)
    assert rc == 0, lib.ocm_last_error()
torch.cuda.synchronize()
nwg = 128
buf = np.zeros(nwg * 128, dtype=np.uint64); raw.ocm_debug_wstamps_linear(buf.ctypes.data_as(C.c_void_p), nwg * 128)
s = buf.reshape(nwg, 16, 8).astype(np.int64)
nw = 12 if s[0, 11, 2] else 8 if s[0, 7, 2] else 4
base = s[:, :nw, 2].min(axis=1, keepdims=True)  # first wave out of the barrier
names = ["enter", "dma landed", "barrier out", "dma issued", "last mfma done"]
print(f"variant {v} ({M}x{N}x{K}, epilogue {epi}): median cycles relative to the first wave leaving the barrier, per wave")
for k in range(5):
    rel = np.median(s[:, :nw, k] - base, axis=0).astype(int)
    print(f"  {names[k]:15s}", " ".join(f"{x:6d}" for x in rel))
