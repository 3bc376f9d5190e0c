#!/usr/bin/env python3
"""Development tool (GPU box): times the split-bf16 (and bf16) nn.Linear kernel variants selectable through
ocm_debug_knob(0, v) at the bench shapes (ViT-S/16, B=64: M = 12608) and checks every variant against variant 0.
    python tools/microbench_x3.py [--prec 2] [--variants 0,1,2,3,4,5,6] [--iters 30]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the knobs exist in the development build only (`make -C vit-ocm-wmsegmentation_amd/csrc dev`)
os.environ.setdefault("OCM_VIT_LIB", os.path.join(ROOT, "exp_libs", "libocm_vit_dev.so"))
import torch  # noqa: E402

from vit_ocm_wmsegmentation_amd import _lib  # noqa: E402
from vit_ocm_wmsegmentation_amd.engine import to_operand  # noqa: E402


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


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
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", type=int, default=2)
    ap.add_argument("--variants", default="0,1,2,3,4,5,6")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rows", type=int, default=12608)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    M, D = a.rows, a.dim
    shapes = {"fc1": (M, 4 * D, D, 2), "fc2": (M, D, 4 * D, 1), "proj": (M, D, D, 1), "qkv_as_linear": (M, 3 * D, D, 3)}
    only = set(filter(None, a.only.split(",")))
    for name, (m, n, k, epi) in shapes.items():
        if only and name not in only:
            continue
        x = torch.randn((m, k), generator=g).to(dev)
        w = (torch.randn((n, k), generator=g) * 0.02).to(dev)
        bias = torch.randn((n,), generator=g).to(dev)
        xa, wa = to_operand(x, a.prec), to_operand(w, a.prec)
        act_out = epi in (2, 3)
        odt = torch.float32 if not act_out else {0: torch.bfloat16, 1: torch.float32, 2: torch.int32}[a.prec]
        resid = torch.randn((m, n), generator=g).to(dev)
        ref = None
        for v in map(int, a.variants.split(",")):
            _lib.check(lib.ocm_debug_knob(0, v))
            out = resid.clone() if epi == 1 else torch.zeros((m, n), dtype=odt, device=dev)

            def fn():
                return lib.ocm_op_linear(a.prec, p(xa), p(wa), p(bias), p(out) if epi == 1 else None, p(out), m, n, k,
                                         3 if epi == 1 else epi, s())  # timing: no in-place accumulation drift
            if epi == 1:
                rc = lib.ocm_op_linear(a.prec, p(xa), p(wa), p(bias), p(out), p(out), m, n, k, 1, s())
            else:
                rc = fn()
            assert rc == 0, lib.ocm_last_error()
            torch.cuda.synchronize()
            got = out.clone()
            if ref is None:
                ref = got
            same = torch.equal(got.view(torch.int32) if got.dtype != torch.bfloat16 else got.view(torch.int16),
                               ref.view(torch.int32) if ref.dtype != torch.bfloat16 else ref.view(torch.int16))
            if epi == 1:  # time the fp32-out epilogue 0 instead of accumulating in place
                o2 = torch.empty((m, n), dtype=torch.float32, device=dev)
                us = timeit(lambda: lib.ocm_op_linear(a.prec, p(xa), p(wa), p(bias), p(resid), p(o2), m, n, k, 1, s()), a.iters)
            else:
                us = timeit(fn, a.iters)
            fl = 2.0 * m * n * k
            print(f"{name:14s} variant {v}: {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s (algorithmic)  bit-identical to v0: {same}")
    _lib.check(lib.ocm_debug_knob(0, 0))
    if not only or "qkv" in only:  # the real qkv projection (head-major q / k / V^T scatter epilogues), knob 3
        B, N, H = M // 197, 197, D // 64
        x = torch.randn((B * N, D), generator=g).to(dev)
        w = (torch.randn((3 * D, D), generator=g) * 0.02).to(dev)
        bias = torch.randn((3 * D,), generator=g).to(dev)
        xa, wa = to_operand(x, a.prec), to_operand(w, a.prec)
        npad = lib.ocm_n_pad_prec(a.prec, N)
        edt = {0: torch.bfloat16, 1: torch.float32, 2: torch.int32}[a.prec]
        ref = None
        for v in (-1, 0, 1, 2, 3):
            _lib.check(lib.ocm_debug_knob(3, v))
            q = torch.zeros((B * H, npad, 64), dtype=edt, device=dev)
            k = torch.zeros_like(q)
            vt = torch.zeros((B * H, 64, npad), dtype=edt, device=dev)

            def fn():
                return lib.ocm_op_qkv_proj(a.prec, p(xa), p(wa), p(bias), p(q), p(k), p(vt), None, B, N, H, s())
            assert fn() == 0, lib.ocm_last_error()
            torch.cuda.synchronize()
            got = torch.cat([q.flatten().view(torch.int16), k.flatten().view(torch.int16), vt.flatten().view(torch.int16)])
            if ref is None:
                ref = got
            us = timeit(fn, a.iters)
            print(f"qkv (B={B})     knob3 {v:2d}: {us:8.2f} us  {2.0 * B * N * D * 3 * D / us / 1e6:7.1f} TFLOP/s (algorithmic)  "
                  f"bit-identical to the register-staged kernel: {torch.equal(got, ref)}")
        _lib.check(lib.ocm_debug_knob(3, 0))


if __name__ == "__main__":
    main()
