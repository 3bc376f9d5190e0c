#!/usr/bin/env python3
"""Development tool (GPU box, development library): A/B of the full-row GEMM + residual + LayerNorm kernel variants
(dev knob 4: 0 = shipped, 1.. = csrc/gemm_kernels.h::launch_resid_ln_d) at the bench shapes (ViT-S/16, B = 64: M = 12 608,
D = 384, K = 384 / 1536). Interleaved rounds in one process, outputs compared bit for bit with variant 0.
    python tools/mb_fused.py [--variants 0,2,3] [--rounds 5] [--iters 20] [--prec 2]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OCM_VIT_LIB", os.path.join(ROOT, "exp_libs", "libocm_vit_dev.so"))
import torch  # noqa: E402

from vit_ocm_wmsegmentation_amd import _lib  # noqa: E402
from vit_ocm_wmsegmentation_amd.engine import to_operand  # noqa: E402


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,2,3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--prec", type=int, default=2)
    ap.add_argument("--M", type=int, default=12608)
    ap.add_argument("--D", type=int, default=384)
    ap.add_argument("--asets", type=int, default=1, help="A operand copies cycled through (> 1: A comes from beyond L2 / the Infinity Cache)")
    ap.add_argument("--wsets", type=int, default=1, help="W operand copies cycled through")
    ap.add_argument("--ks", default="", help="comma list of K (default D,4D)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    variants = [int(v) for v in args.variants.split(",")]
    M, D = args.M, args.D
    act = {0: torch.bfloat16, 1: torch.float32, 2: torch.int32}[args.prec]
    g = torch.Generator().manual_seed(0)
    for K in ([int(k) for k in args.ks.split(",")] if args.ks else (D, 4 * D)):
        a = torch.randn((M, K), generator=g).to(dev)
        w = (torch.randn((D, K), generator=g) * 0.05).to(dev)
        bias = torch.randn((D,), generator=g).to(dev) * 0.1
        resid = torch.randn((M, D), generator=g).to(dev)
        gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        a_s, w_s = to_operand(a, args.prec), to_operand(w, args.prec)
        a_sets = [a_s] + [a_s.clone() for _ in range(args.asets - 1)]
        w_sets = [w_s] + [w_s.clone() for _ in range(args.wsets - 1)]
        cnt = [0]
        # a second, cold-ish operand set so that back-to-back launches do not find everything in L2
        outs = {}
        times = {v: [] for v in variants}

        def run(v, x, xn):
            _lib.check(lib.ocm_debug_knob(4, v))
            cnt[0] += 1
            a_c, w_c = a_sets[cnt[0] % args.asets], w_sets[cnt[0] % args.wsets]
            _lib.check(lib.ocm_op_linear_resid_ln(args.prec, p(a_c), p(w_c), p(bias), p(resid), p(x), p(gamma), p(beta), p(xn),
                                                  M, D, K, 1e-6, st()))
        for v in variants:
            x = torch.empty((M, D), device=dev)
            xn = torch.empty((M, D), dtype=act, device=dev)
            run(v, x, xn)
            torch.cuda.synchronize()
            outs[v] = (x, xn)
        for v in variants[1:]:
            same = torch.equal(outs[v][0], outs[variants[0]][0]) and torch.equal(outs[v][1].view(torch.int32) if act != torch.bfloat16 else outs[v][1].view(torch.int16),
                                                                                 outs[variants[0]][1].view(torch.int32) if act != torch.bfloat16 else outs[variants[0]][1].view(torch.int16))
            print(f"K={K}: variant {v} bit-identical to variant {variants[0]}: {same}")
        for _ in range(args.rounds):
            for v in variants:
                x, xn = outs[v]
                for _ in range(3):
                    run(v, x, xn)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run(v, x, xn)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.iters * 1e3)
        for v in variants:
            t = sorted(times[v])
            fl = 2.0 * M * D * K
            print(f"K={K:5d} variant {v}: median {t[len(t) // 2]:7.2f} us  min {t[0]:7.2f} us   {fl / t[len(t) // 2] / 1e6:7.1f} TFLOP/s algorithmic")
    lib.ocm_debug_knob(4, 0)


if __name__ == "__main__":
    main()
