#!/usr/bin/env python3
"""Experiment: does running two half-batches on two HIP streams overlap kernel tails/prologues?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
from vit_ocm_wmsegmentation_amd import _lib, synth

dev = torch.device("cuda:0")
def mk():
    m = vits.vit_small(patch_size=16, num_classes=0)
    m.load_state_dict(synth.synth_arch_state_dict("vit_small", 16, variant="init"))
    return m.eval().to(dev)
flags = _lib.OCM_OUT_ATTN | _lib.OCM_LAST_ATTN_ONLY
x = synth.synth_tiles(64, 224).to(dev)
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
m0 = mk()
print("1 stream  B=64:", round(bench(lambda: m0._run(x, flags=flags)), 3), "ms")
for nsplit in (2, 4):
    models = [mk() for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = x.chunk(nsplit)
    def run():
        for m, s, xi in zip(models, streams, xs):
            with torch.cuda.stream(s):
                m._run(xi, flags=flags)
    print(f"{nsplit} streams B={64 // nsplit} each:", round(bench(run), 3), "ms")
