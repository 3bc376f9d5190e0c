"""Mirror of the reference's `dino` package: `dino.vision_transformer` keeps the module surface
(`vit_tiny/small/base`, `VisionTransformer` and its methods, identical state_dict keys) that
eval.py / sw_processing.py / analyse_attention.py / PGT.py / model.py bind to."""
