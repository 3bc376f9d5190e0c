#!/usr/bin/env python3
"""Swin-T throughput (BASELINE config 5: 224^2, batch 256, one MI355X). Development tool, GPU box only.

    python tools/bench_swin.py [--batch 256] [--steps 10] [--precision bf16x3]

Prints one human line and one JSON line. The JSON's `roofline` is that of the DOMINANT KERNEL CLASS by device time, measured live
with HIP events on the launch stream (ocm_prof_begin / ocm_prof_end brackets every launch of the Swin engine), with the class's
own algorithmic FLOPs and bytes per forward:
  * window attention (swin_wattn_*): 4 T 49 C FLOPs per layer (scores + context over 7 x 7 windows) and q, k, v in / context out
    = 4 T C E bytes (E = bytes per operand element: 4 for split-bf16 pairs and fp32, 2 for bf16) — byte-bound by two orders of
    magnitude (24 FLOP per byte), so its roofline is the HBM one; in split-bf16 precision the layers of 96 / 192 channels run
    their attention half as one kernel (swin_attn_block_x3_kernel) counted in this class with the projections' FLOPs;
  * the GEMM classes: 2 T K N FLOPs against the dense MFMA peak of the mode.
`kernel_breakdown` lists every class; `profile` names the rocprofv3 summary of the same command committed for this round.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vit_ocm_wmsegmentation_amd import _lib  # noqa: E402
from vit_ocm_wmsegmentation_amd import swin as SW  # noqa: E402
from vit_ocm_wmsegmentation_amd import synth  # noqa: E402

PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def class_work(batch, cfg, esz, fused_attn=False):
    """Algorithmic (FLOPs, bytes) per forward of each kernel class of the Swin engine (swin_engine.hip's PROF classes).
    fused_attn (split-bf16 precision, round 4): the layers of 96 channels run layernorm_before, q | k | v, window attention, o_proj
    and the residual as ONE kernel profiled under `attention` (x read twice, written once), those of 192 channels everything up
    to the context pairs (x read once, context written); their FLOPs and bytes move to that class."""
    side = cfg["image_size"] // cfg["patch_size"]
    C0, ws, ratio = cfg["embed_dim"], cfg["window_size"], cfg["mlp_ratio"]
    work = {k: [0.0, 0.0] for k in ("patch_embed", "qkv_gemm", "attention", "proj_gemm", "fc1_gemm", "fc2_gemm")}
    T = batch * side * side
    work["patch_embed"] = [2.0 * T * cfg["num_channels"] * cfg["patch_size"] ** 2 * C0,
                           4.0 * batch * cfg["num_channels"] * cfg["image_size"] ** 2 + 4.0 * T * C0]
    for st, depth in enumerate(cfg["depths"]):
        Cs = C0 << st
        M = int(ratio * Cs)
        for _ in range(depth):
            if fused_attn and Cs in (96, 192):
                work["attention"][0] += 4.0 * T * ws * ws * Cs + 2.0 * T * Cs * 3 * Cs
                if Cs == 96:  # the whole half in one kernel
                    work["attention"][0] += 2.0 * T * Cs * Cs
                    work["attention"][1] += 12.0 * T * Cs
                else:  # up to the context pairs; o_proj stays a GEMM
                    work["attention"][1] += 4.0 * T * Cs + esz * T * Cs
                    work["proj_gemm"][0] += 2.0 * T * Cs * Cs
                    work["proj_gemm"][1] += esz * T * Cs + 8.0 * T * Cs
            else:
                work["qkv_gemm"][0] += 2.0 * T * Cs * 3 * Cs
                work["qkv_gemm"][1] += 4.0 * T * Cs + esz * T * 3 * Cs
                work["attention"][0] += 4.0 * T * ws * ws * Cs
                work["attention"][1] += esz * T * 4 * Cs
                work["proj_gemm"][0] += 2.0 * T * Cs * Cs
                work["proj_gemm"][1] += esz * T * Cs + 8.0 * T * Cs
            work["fc1_gemm"][0] += 2.0 * T * Cs * M
            work["fc1_gemm"][1] += 4.0 * T * Cs + esz * T * M
            work["fc2_gemm"][0] += 2.0 * T * M * Cs
            work["fc2_gemm"][1] += esz * T * M + 8.0 * T * Cs
        if st + 1 < len(cfg["depths"]):
            work["proj_gemm"][0] += 2.0 * (T // 4) * 4 * Cs * 2 * Cs  # patch-merging reduction
            work["proj_gemm"][1] += esz * (T // 4) * 4 * Cs + 4.0 * (T // 4) * 2 * Cs
            T //= 4
    return work


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--profile", default="profiles/r04_kernel_stats_swin.csv",
                    help="rocprofv3 --kernel-trace --stats summary of this command committed for the round")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = SW.SwinConfig(num_labels=5)
    model = SW.SwinForImageClassification(cfg)
    model.load_state_dict(synth.synth_swin_state_dict(synth.SWIN_TINY, seed=1))
    model = model.to(dev).eval().set_precision(a.precision)
    x = synth.synth_tiles(a.batch, 224, seed=3).to(dev)
    for _ in range(3):
        model(pixel_values=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = model(pixel_values=x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flops = 4.5e9 * a.batch  # ~4.5 GFLOP per 224^2 image (Swin-T)
    print(f"swin-tiny 224^2 batch {a.batch} {a.precision}: {dt * 1e3:.2f} ms/step = {a.batch / dt:.0f} images/s "
          f"(~{flops / dt / 1e12:.0f} TFLOP/s); logits[0] = {out.logits[0].cpu().numpy().round(3)}")

    # per-class device time (HIP events around every launch of the engine, a separate untimed pass)
    lib = _lib.load()
    ncls = len(_lib.KERNEL_CLASSES)
    ms, cnt = (C.c_double * ncls)(), (C.c_int64 * ncls)()
    reps = 3
    _lib.check(lib.ocm_prof_begin(0xFFFFFFFF, reps * 256))
    for _ in range(reps):
        model(pixel_values=x)
    torch.cuda.synchronize()
    _lib.check(lib.ocm_prof_end(ms, cnt))
    esz = 2 if a.precision == "bf16" else 4
    mpp = 3 if a.precision == "bf16x3" else 1
    peak_mfma = 157.3 if a.precision == "fp32" else 2500.0  # exact-fp32 MFMA / dense bf16 MFMA
    work = class_work(a.batch, synth.SWIN_TINY, esz, fused_attn=a.precision == "bf16x3")
    breakdown = {}
    for i, name in enumerate(_lib.KERNEL_CLASSES):
        if cnt[i]:
            per_fwd = ms[i] / reps
            breakdown[name] = {"launches_per_fwd": cnt[i] // reps, "ms_per_fwd": round(per_fwd, 3)}
            if name in work:
                breakdown[name]["tflops"] = round(work[name][0] / (per_fwd * 1e-3) / 1e12, 1)
                breakdown[name]["gbs"] = round(work[name][1] / (per_fwd * 1e-3) / 1e9, 0)
    dom = max((n for n in breakdown if n in work), key=lambda n: breakdown[n]["ms_per_fwd"])
    t_dom = breakdown[dom]["ms_per_fwd"] * 1e-3
    f_frac = work[dom][0] / t_dom / 1e12 / peak_mfma
    b_frac = work[dom][1] / t_dom / 1e9 / PEAK_HBM_GBS
    if b_frac >= f_frac:
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(work[dom][1] / t_dom / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(b_frac, 4), "algorithmic_bytes_per_fwd": work[dom][1], "flops_per_fwd": work[dom][0],
                "mfma_frac": round(f_frac, 4)}
    else:
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(work[dom][0] / t_dom / 1e12, 1), "peak": peak_mfma, "unit": "TFLOP/s",
                "frac": round(f_frac, 4), "flops_per_fwd": work[dom][0], "algorithmic_bytes_per_fwd": work[dom][1],
                "hbm_frac": round(b_frac, 4), "mfma_per_product": mpp, "mfma_pipe_frac": round(f_frac * mpp, 4)}
    roof.update({"ms_per_fwd": breakdown[dom]["ms_per_fwd"], "launches_per_fwd": breakdown[dom]["launches_per_fwd"], "traffic": None,
                 "profile": a.profile})
    print(json.dumps({"metric": "Swin-T images/s (224x224, BASELINE config 5)", "value": round(a.batch / dt, 1), "unit": "images/s",
                      "ms_per_step": round(dt * 1e3, 3), "dtype": a.precision, "config": {"workload": f"swin-tiny 224^2, batch {a.batch}"},
                      "path_tflops": round(flops / dt / 1e12, 1), "path_frac_of_mfma_peak": round(flops / dt / 1e12 / peak_mfma, 4),
                      "roofline": roof, "kernel_breakdown": breakdown}))


if __name__ == "__main__":
    main()
