"""MI355X-native ViT attention-map hot path of linum-uqam/ViT-OCM-WMSegmentation.

Layout
  csrc/                  hand-written HIP kernels (gfx950) + the C ABI (include/ocm_vit.h)
  _lib.py                ctypes binding of libocm_vit.so
  engine.py              device-pointer plumbing between torch tensors and the C ABI
  dino/vision_transformer.py   nn.Module mirror of the reference's module (drop-in surface)
  utils.py               compute_attention / region-query index (reference utils.py:229-235)
  sw_processing.py       sliding-window tile sharding across GPUs + RCCL all-gather
  synth.py               deterministic synthetic weights / tiles (no checkpoints exist offline)
"""
__version__ = "0.1.0"
