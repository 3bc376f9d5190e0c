#!/usr/bin/env python3
"""Development tool (GPU box): does running the bench batch as two half batches on two HIP streams (two engine handles,
the launch gaps and tile-quantisation tails of one stream filled by the other stream's kernels) beat one batch on one stream?
    python tools/two_stream.py [batch] [steps] [splits]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits  # noqa: E402
from vit_ocm_wmsegmentation_amd import _lib, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
splits = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4]
dev = torch.device("cuda:0")
flags = _lib.OCM_OUT_ATTN | _lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY
sd = synth.synth_arch_state_dict("vit_small", 16, seed=0, variant="init")
x = synth.synth_tiles(B, 224, seed=1234).to(dev)


def make():
    m = vits.vit_small(patch_size=16, num_classes=0)
    m.load_state_dict(sd)
    for q in m.parameters():
        q.requires_grad = False
    return m.eval().to(dev)


ref = None
for n in splits:
    models = [make() for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    bounds = [B * i // n for i in range(n + 1)]
    parts = [x[bounds[i]:bounds[i + 1]].contiguous() for i in range(n)]

    def step():
        outs = []
        for m, s, xp in zip(models, streams, parts):
            with torch.cuda.stream(s):
                outs.append(m._run(xp, flags=flags))
        return outs

    for _ in range(5):
        outs = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    attn = torch.cat([o["attn"][0] for o in outs])
    if ref is None:
        ref = attn.clone()
    print(f"{n} stream(s) x {B // n} tiles: {dt * 1e3:.3f} ms per {B} tiles = {B / dt:.0f} tiles/s; "
          f"max |attn - single stream| = {float((attn - ref).abs().max()):.2e}", flush=True)
    del models, outs
