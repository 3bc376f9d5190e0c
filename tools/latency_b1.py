#!/usr/bin/env python3
"""Single-tile latency of the drop-in calls (what eval.py / sw_processing.py do per image: B = 1).
Development tool, GPU box only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits  # noqa: E402
from vit_ocm_wmsegmentation_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
for arch, p, S in (("vit_small", 16, 224), ("vit_small", 8, 384)):
    model = vits.__dict__[arch](patch_size=p, num_classes=0)
    model.load_state_dict(synth.synth_arch_state_dict(arch, p, seed=0, variant="init"))
    model = model.eval().to(dev).set_precision(PREC)
    x = synth.synth_tiles(1, S, seed=1).to(dev)
    for name, fn in (("get_intermediate_feat", lambda: model.get_intermediate_feat(x, n=1)),
                     ("get_last_selfattention", lambda: model.get_last_selfattention(x)),
                     ("get_last_attention_rows", lambda: model.get_last_attention_rows(x))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            fn()
            torch.cuda.synchronize()  # the reference's callers read the result back after every call
        dt = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dq = (time.perf_counter() - t1) / n
        line = f"[{PREC}] {arch}/{p} {S}^2 B=1 {name:24s}: {dt * 1e3:7.3f} ms per call (synced), {dq * 1e3:7.3f} ms queued back-to-back"
        if name != "get_intermediate_feat":
            run = model.graphed(name)  # HIP-graph replay of the same call
            for _ in range(3):
                run(x)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(n):
                run(x)
                torch.cuda.synchronize()
            line += f", {(time.perf_counter() - t2) / n * 1e3:7.3f} ms as a graph replay (synced)"
        print(line, flush=True)
