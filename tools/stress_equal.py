#!/usr/bin/env python3
"""Development: run-to-run determinism of the ViT-S/16 B = 64 forward. `get_last_selfattention` must return the same bits as the
last block's attention of `get_intermediate_feat`, call after call (a race in any kernel of the chain shows here as a mismatch).

    python tools/stress_equal.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
from vit_ocm_wmsegmentation_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
model = vits.vit_small(patch_size=16, num_classes=0)
model.load_state_dict(synth.synth_arch_state_dict("vit_small", 16, variant="init"))
model = model.eval().to(dev)
x = synth.synth_tiles(64, 224, seed=1234).to(dev)
ref = model.get_last_selfattention(x)
bad = 0
for i in range(rounds):
    a = model.get_last_selfattention(x)
    if len(sys.argv) > 2:  # any second argument: other batch sizes and entry points in between (workspace switches, as the tests do)
        model.get_last_attention_rows(x)
        model.get_last_selfattention(x[31:32])
        junk = torch.full((int(sys.argv[2]) * 1024 * 256,), float("nan"), device=dev)  # recycle allocator blocks with NaNs
        del junk
        model.get_last_selfattention(x[:7])
    f, attns, q = model.get_intermediate_feat(x, n=1)
    e1, e2 = torch.equal(a, ref), torch.equal(attns[0], ref)
    if not (e1 and e2):
        bad += 1
        d1 = (a - ref).abs().max().item()
        d2 = (attns[0] - ref).abs().max().item()
        nz = ((attns[0] != ref).flatten(1).any(1) | (a != ref).flatten(1).any(1)).nonzero().flatten().tolist()
        print(f"round {i}: last_selfattention equal {e1} ({d1:.2e}), intermediate_feat equal {e2} ({d2:.2e}); images {nz[:10]}")
print(f"{bad} mismatching rounds of {rounds}")
