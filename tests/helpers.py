"""Shared test helpers: rebuild a golden case's weights / inputs / oracle config from its seeds."""
import os

import numpy as np

from tests.golden_cases import CASES
from vit_ocm_wmsegmentation_amd import synth

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_dims(case):
    if "arch" in case:
        return synth.ARCHS[case["arch"]]
    return case["dim"], case["depth"], case["heads"]


def case_state_dict(case):
    D, L, _ = case_dims(case)
    return synth.synth_state_dict(D, L, case["patch"], seed=case["seed"], variant=case["variant"],
                                  img_size=case["img_size"])


def case_inputs(case):
    return [synth.synth_tiles(B, H, W, seed=s) for (B, H, W, s) in case["inputs"]]


def load_golden(name):
    return np.load(os.path.join(GOLD_DIR, name + ".npz"))


def build_module(case, device):
    """The product's nn.Module for a golden case, weights loaded through load_state_dict."""
    from functools import partial

    import torch.nn as nn

    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    D, L, H = case_dims(case)
    if "arch" in case:
        model = vits.__dict__[case["arch"]](patch_size=case["patch"], num_classes=0)
    else:
        model = vits.VisionTransformer(img_size=[case["img_size"]], patch_size=case["patch"], embed_dim=D, depth=L,
                                       num_heads=H, mlp_ratio=4, qkv_bias=True,
                                       norm_layer=partial(nn.LayerNorm, eps=1e-6), num_classes=0)
    msg = model.load_state_dict(case_state_dict(case), strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    for p in model.parameters():
        p.requires_grad = False
    return model.eval().to(device)


__all__ = ["CASES", "case_dims", "case_state_dict", "case_inputs", "load_golden", "build_module"]
