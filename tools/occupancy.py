#!/usr/bin/env python3
"""Workgroups per CU the runtime grants the main split-bf16 kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor).
Development tool, GPU box only; needs the stamps build: `make -C vit-ocm-wmsegmentation_amd/csrc stamps`."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
os.environ["OCM_VIT_LIB"] = os.path.join(os.getcwd(), "exp_libs", "stamps.so")
import torch
from vit_ocm_wmsegmentation_amd import _lib
lib = _lib.load(); raw = C.CDLL(os.environ["OCM_VIT_LIB"])
torch.zeros(1, device="cuda")
out = (C.c_int * 8)()
n = raw.ocm_debug_occupancy(out, 8)
print("workgroups per CU: fc2 64x128 dma/2, fc1 128x128 dma/2, qkv 128x128q dma/2, fused 64x384:", list(out)[:n])
