#!/usr/bin/env python3
"""Development tool (GPU box; `make -C vit-ocm-wmsegmentation_amd/csrc stamps`): per-wave timeline of the split-bf16
streaming attention kernel (attn_fwd_x3_dma_kernel) at a bench shape.
    python tools/attn_stamps.py [B N H]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ["OCM_VIT_LIB"] = os.path.join(ROOT, "exp_libs", "stamps.so")
from vit_ocm_wmsegmentation_amd import _lib  # noqa: E402
from vit_ocm_wmsegmentation_amd.engine import to_operand  # noqa: E402

lib = _lib.load()
raw = C.CDLL(os.environ["OCM_VIT_LIB"])
dev = torch.device("cuda:0")
B, N, H = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 197, 6)
D = H * 64
g = torch.Generator().manual_seed(0)
x = to_operand(torch.randn((B * N, D), generator=g).to(dev), 2)
w = to_operand((torch.randn((3 * D, D), generator=g) * 0.05).to(dev), 2)
bias = torch.zeros(3 * D, device=dev)
npad = lib.ocm_n_pad_prec(2, N)
q = torch.zeros((B * H, npad, 64), dtype=torch.int32, device=dev)
k = torch.zeros_like(q)
vt = torch.zeros((B * H, 64, npad), dtype=torch.int32, device=dev)
ctx = torch.zeros((B * N, D), dtype=torch.int32, device=dev)
lse = torch.zeros((B * H, N), dtype=torch.float32, device=dev)


def p(t):
    return C.c_void_p(t.data_ptr())


st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.ocm_op_qkv_proj(2, p(x), p(w), p(bias), p(q), p(k), p(vt), None, B, N, H, st))
for _ in range(3):
    _lib.check(lib.ocm_op_attention(2, p(q), p(k), p(vt), p(ctx), p(lse), B, N, H, 0.125, st))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    lib.ocm_op_attention(2, p(q), p(k), p(vt), p(ctx), p(lse), B, N, H, 0.125, st)
b.record()
torch.cuda.synchronize()
print(f"attention B={B} N={N} H={H}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch (stand-alone, stamped build)")

nwg = min(1024, ((N + 31) // 32 + 3) // 4 * B * H)
buf = np.zeros(nwg * 8 * 16, dtype=np.uint64)
raw.ocm_debug_stamps_attn(buf.ctypes.data_as(C.c_void_p), buf.size)
s = buf.reshape(nwg, 8, 16).astype(np.int64)[:, :4]
ntiles = (N + 31) // 32
# points 12 / 13: the constant 100 MHz counter at entry / exit (one time base for the whole device)
rt0 = s[:, 0, 12].min()
ent, ext = (s[:, 0, 12] - rt0) * 10, (s[:, 0, 13] - rt0) * 10  # ns
print(f"entry of the workgroups, ns after the first: median {int(np.median(ent))} p90 {int(np.percentile(ent, 90))} max {int(ent.max())}")
print(f"exit of the workgroups, ns after the first entry: min {int(ext.min())} median {int(np.median(ext))} max {int(ext.max())}")
life_ns = (s[:, 0, 13] - s[:, 0, 12]) * 10
life_cy = s[:, 0, 15] - s[:, 0, 0]
print(f"workgroup lifetime: median {int(np.median(life_ns))} ns = {int(np.median(life_cy))} cycles -> "
      f"{np.median(life_cy) / max(np.median(life_ns), 1):.2f} GHz shader clock")
rel = s[:, 0, :] - s[:, 0, 0:1]
names = ["entered", "Q + 2 tiles landed"] + [f"barrier tile {i}" for i in range(min(ntiles, 12))] + ["loop done", "stores issued"]
cols = list(range(2 + min(ntiles, 12))) + [14, 15]
prev = 0
for name, c in zip(names, cols):
    med = int(np.median(rel[:, c]))
    print(f"  {name:22s} median {med:8d} cycles after the workgroup's entry (+{med - prev})")
    prev = med
# by order of entry: do late workgroups live shorter lives (less contention)?
o = np.argsort(ent)
for lo, hi in ((0, nwg // 4), (nwg // 4, nwg // 2), (nwg // 2, 3 * nwg // 4), (3 * nwg // 4, nwg)):
    sel = o[lo:hi]
    print(f"  workgroups {lo}-{hi} by entry: enter at {int(np.median(ent[sel]))} ns, live {int(np.median(life_ns[sel]))} ns")
