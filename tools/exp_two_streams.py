#!/usr/bin/env python3
"""Experiment (GPU box): does running the forward as two half-batches on two HIP streams fill the idle tail rounds of the
per-kernel grids? Two model copies (own engines / workspaces), 32 tiles each, vs one model on 64 tiles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
from vit_ocm_wmsegmentation_amd import _lib, synth

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = 64
sd = synth.synth_arch_state_dict("vit_small", 16, seed=0, variant="init")
models = []
for _ in range(4):
    m = vits.vit_small(patch_size=16, num_classes=0)
    m.load_state_dict(sd)
    models.append(m.eval().to(dev).set_precision(prec))
x = synth.synth_tiles(B, 224, seed=1234).to(dev)
flags = _lib.OCM_OUT_ATTN | _lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY


def run(parts, steps=20):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = x.chunk(parts)

    def step():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        outs = []
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs.append(models[i]._run(chunks[i], flags=flags))
        for s in streams:
            cur.wait_stream(s)
        return outs
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt


for parts in (1, 2, 4, 1, 2, 4):
    dt = run(parts)
    print(f"{prec}: {parts} stream(s) x {B // parts} tiles: {dt * 1e3:.3f} ms/step  {B / dt:.0f} tiles/s", flush=True)
