"""The golden-fixture cases: shared by oracle/make_golden.py (which writes tests/golden/<name>.npz from
the real reference) and by the tests (which regenerate weights/inputs from the same seeds).
  arch | (dim, depth, heads): model;  patch, img_size: constructor;  variant, seed: synth.synth_state_dict;
  inputs: [(batch, height, width, tile_seed)];  n: last-n blocks returned;  full: store complete tensors."""
CASES = {
    # name: dict(ctor=..., heads, patch, img_size, variant, seed, inputs=[(B, H, W, seed)], n)
    "tiny_p8": dict(dim=128, depth=2, heads=2, patch=8, img_size=32, variant="full", seed=1,
                    inputs=[(2, 32, 32, 11), (1, 48, 32, 12)], n=2, full=True),
    "vits16_full": dict(arch="vit_small", patch=16, img_size=224, variant="full", seed=0,
                        inputs=[(2, 224, 224, 1234)], n=1),
    "vits16_sharp": dict(arch="vit_small", patch=16, img_size=224, variant="sharp", seed=0,
                         inputs=[(2, 224, 224, 1234)], n=1),
    "vits16_peaked": dict(arch="vit_small", patch=16, img_size=224, variant="peaked", seed=0,
                          inputs=[(2, 224, 224, 1234)], n=1),
    "vits16_init": dict(arch="vit_small", patch=16, img_size=224, variant="init", seed=0,
                        inputs=[(1, 224, 224, 1234)], n=1),
    "vitt16_full": dict(arch="vit_tiny", patch=16, img_size=224, variant="full", seed=3,
                        inputs=[(1, 224, 224, 77)], n=1),
    "vits8_384_sharp": dict(arch="vit_small", patch=8, img_size=224, variant="sharp", seed=0,
                            inputs=[(1, 384, 384, 4321)], n=1),
    "vitb16_384_full": dict(arch="vit_base", patch=16, img_size=224, variant="full", seed=0,
                            inputs=[(1, 384, 384, 99)], n=1),
}


# model.py wrappers: encoder geometry + decoder stride; weights / masks from synth.py.
# img_size != 224 exercises the interpolated-position branch (model.py:38-39), 224 the native one.
WRAPPER_CASES = {
    "wrap_p8_64": dict(dim=128, depth=2, heads=2, patch=8, img_size=64, batch=2, seed=11, variant="full"),
    "wrap_p16_224": dict(dim=128, depth=2, heads=2, patch=16, img_size=224, batch=1, seed=12, variant="sharp"),
    # the reference's own build_model() encoder (model.py:93-103): depth 4, THREE heads of 128 channels
    "wrap_mim_hd128": dict(dim=384, depth=4, heads=3, patch=8, img_size=64, batch=2, seed=13, variant="sharp"),
}


# Swin: fixtures come from the installed transformers package (oracle/make_golden_swin.py).
# "tiny224" is the reference's configuration (SwinConfig defaults, num_labels=5, ADB/train.py:70-77);
# "small56" is a 2-stage miniature (image 56 -> 14x14 -> 7x7 grids) that exercises shift masks and the
# window-equals-grid clamp in seconds.
SWIN_CASES = {
    "tiny224": dict(batch=2, seed=21, qk_gain=6.0),
    "small56": dict(batch=3, seed=22, qk_gain=6.0, cfg=dict(image_size=56, depths=(2, 2), num_heads=(3, 6))),
}
