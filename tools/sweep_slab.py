#!/usr/bin/env python3
"""Sliding-window sweep of a synthetic slab (BASELINE.json configs[3]: ViT-S/8, 384-px windows at stride 128)
through SlidingWindowAttention on the local rank(s). Prints windows/s. GPU box only.

    python tools/sweep_slab.py [--size 4096] [--batch auto|16] [--arch vit_small] [--patch 8]
    python -m torch.distributed.run --nproc-per-node N ... tools/sweep_slab.py   (tile shard + all-gather)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
from vit_ocm_wmsegmentation_amd import synth
from vit_ocm_wmsegmentation_amd.sw_processing import SlidingWindowAttention

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--batch", default="auto", help="windows per forward, or auto (SlidingWindowAttention.auto_batch_plan)")
ap.add_argument("--arch", default="vit_small")
ap.add_argument("--patch", type=int, default=8)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
model = vits.__dict__[a.arch](patch_size=a.patch, num_classes=0)
model.load_state_dict(synth.synth_arch_state_dict(a.arch, a.patch, variant="init"))
model.eval().to(dev)
slab = synth.synth_tiles(1, a.size, seed=7)[0].to(dev)
sweep = SlidingWindowAttention(model, window=384, stride=128, batch_tiles=a.batch if a.batch == "auto" else int(a.batch))
maps = sweep(slab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    maps = sweep(slab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
if rank == 0:
    T = maps.shape[0]
    print(f"slab {a.size}^2 -> {T} windows of 384^2 ({a.arch}/{a.patch}, N={(384 // a.patch) ** 2 + 1}), world {world}, "
          f"batch {a.batch}: {dt * 1e3:.1f} ms/sweep = {T / dt:.1f} windows/s; maps {tuple(maps.shape)} "
          f"row-sum of CLS rows in [{float(maps.sum((-1, -2)).min()):.4f}, {float(maps.sum((-1, -2)).max()):.4f}]")
if world > 1:
    dist.destroy_process_group()
