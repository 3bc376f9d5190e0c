#!/usr/bin/env python3
"""Development tool (GPU box): N one-tile get_last_selfattention calls, for `rocprofv3 --kernel-trace --stats`.
    python tools/b1_loop.py [arch patch size calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits  # noqa: E402
from vit_ocm_wmsegmentation_amd import synth  # noqa: E402

arch, p, S, n = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("vit_small", 16, 224, 100)
dev = torch.device("cuda:0")
model = vits.__dict__[arch](patch_size=p, num_classes=0)
model.load_state_dict(synth.synth_arch_state_dict(arch, p, seed=0, variant="init"))
model = model.eval().to(dev)
x = synth.synth_tiles(1, S, seed=1).to(dev)
for _ in range(n):
    model.get_last_selfattention(x)
torch.cuda.synchronize()
