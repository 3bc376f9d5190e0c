#!/usr/bin/env python3
"""Swin-T throughput (BASELINE config 5: 224^2, batch 256, one MI355X). Development tool, GPU box only.

    python tools/bench_swin.py [--batch 256] [--steps 10] [--precision bf16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vit_ocm_wmsegmentation_amd import swin as SW  # noqa: E402
from vit_ocm_wmsegmentation_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="bf16")
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
    import json
    peak = 157.3 if a.precision == "fp32" else 2500.0  # exact-fp32 MFMA / dense bf16 MFMA (split-bf16: 3 MFMAs per product)
    print(json.dumps({"metric": "Swin-T images/s (224x224, BASELINE config 5)", "value": round(a.batch / dt, 1), "unit": "images/s",
                      "ms_per_step": round(dt * 1e3, 3), "dtype": a.precision, "config": {"workload": f"swin-tiny 224^2, batch {a.batch}"},
                      "roofline": {"bound": "mfma", "kernel": "whole forward (GEMMs 60 %, window attention 13 %, LayerNorm 11 %, "
                                   "embedding 6 % of the device time: profiles/r02_kernel_stats_swin.csv)",
                                   "achieved": round(flops / dt / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(flops / dt / 1e12 / peak, 4), "traffic": None,
                                   "mfma_per_product": 3 if a.precision == "bf16x3" else 1}}))


if __name__ == "__main__":
    main()
