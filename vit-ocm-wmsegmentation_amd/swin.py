"""Swin-T inference behind the module surface the reference's Allen_data_Backbone/train.py:70-85 uses
(transformers' `SwinConfig` / `SwinForImageClassification`): same constructor fields, same state_dict keys (a
transformers checkpoint loads with `load_state_dict`), `model(pixel_values=...)` -> object with `.logits`
(and `.pooler_output`, optionally `.last_hidden_state`). The arithmetic runs in libocm_vit.so
(include/ocm_swin.h); there is no CPU fallback. Inference only (BASELINE config 5).
"""
import ctypes as C
import types

import torch
import torch.nn as nn

from . import _lib
from .dino.utils import trunc_normal_
from .engine import _p, _require_hip, _stream


class SwinConfig:
    """The fields of transformers.SwinConfig this path reads (defaults = swin-tiny-patch4-window7-224)."""

    def __init__(self, image_size=224, patch_size=4, num_channels=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, hidden_act="gelu",
                 layer_norm_eps=1e-5, num_labels=2, label2id=None, id2label=None, **unused):
        if hidden_act != "gelu" or not qkv_bias:
            raise ValueError("only hidden_act='gelu' with qkv_bias=True (the Swin-T defaults) is built")
        self.image_size, self.patch_size, self.num_channels, self.embed_dim = image_size, patch_size, num_channels, embed_dim
        self.depths, self.num_heads = tuple(depths), tuple(num_heads)
        self.window_size, self.mlp_ratio, self.layer_norm_eps = window_size, mlp_ratio, layer_norm_eps
        self.qkv_bias, self.hidden_act = qkv_bias, hidden_act
        self.label2id, self.id2label = label2id, id2label
        self.num_labels = len(id2label) if id2label else num_labels
        self.num_layers = len(self.depths)
        self.hidden_size = int(embed_dim * 2 ** (self.num_layers - 1))


def _param_shapes(cfg):
    from .synth import swin_param_shapes
    return swin_param_shapes(dict(image_size=cfg.image_size, patch_size=cfg.patch_size, num_channels=cfg.num_channels,
                                  embed_dim=cfg.embed_dim, depths=cfg.depths, num_heads=cfg.num_heads,
                                  window_size=cfg.window_size, mlp_ratio=cfg.mlp_ratio, num_labels=cfg.num_labels))


class SwinForImageClassification(nn.Module):
    """Parameters are registered flat under their transformers state_dict keys, so `state_dict()` /
    `load_state_dict()` interoperate with transformers checkpoints key for key."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        self._names = {}
        for key, shape in _param_shapes(config).items():
            p = nn.Parameter(torch.zeros(shape), requires_grad=False)
            if "norm" in key and key.endswith("weight"):
                nn.init.ones_(p)
            elif key.endswith("weight") or key.endswith("relative_position_bias_table"):
                trunc_normal_(p, std=.02)
            flat = key.replace(".", "__")
            self.register_parameter(flat, p)
            self._names[flat] = key
        self._register_state_dict_hook(self._rename_out)
        self._register_load_state_dict_pre_hook(self._rename_in)
        self.__dict__["_engine"] = None
        self.__dict__["_precision"] = "bf16"

    # ---- transformers key names in and out ----------------------------------------------------
    @staticmethod
    def _rename_out(module, state_dict, prefix, local_metadata):
        for flat, key in module._names.items():
            if prefix + flat in state_dict:
                state_dict[prefix + key] = state_dict.pop(prefix + flat)
        return state_dict

    def _rename_in(self, state_dict, prefix, *args):
        for flat, key in self._names.items():
            if prefix + key in state_dict:
                state_dict[prefix + flat] = state_dict.pop(prefix + key)

    def set_precision(self, precision):
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        if precision != self._precision:
            self.__dict__["_precision"] = precision
            self._drop_engine()
        return self

    def _drop_engine(self):
        eng = self.__dict__.get("_engine")
        if eng is not None:
            _lib.load().ocm_swin_destroy(eng["h"])
        self.__dict__["_engine"] = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _get_engine(self, device):
        params = [(self._names[n], p) for n, p in self.named_parameters()]
        sig = (device, tuple((p.data_ptr(), p._version) for _, p in params))
        eng = self.__dict__.get("_engine")
        if eng is not None and eng["sig"] == sig:
            return eng
        self._drop_engine()
        lib, c = _lib.load(), self.config
        cfg = _lib.OcmSwinConfig(image_size=c.image_size, patch_size=c.patch_size, num_channels=c.num_channels,
                                 embed_dim=c.embed_dim, num_stages=c.num_layers, window_size=c.window_size,
                                 num_labels=c.num_labels, mlp_ratio=c.mlp_ratio, ln_eps=c.layer_norm_eps,
                                 precision=_lib.PRECISIONS[self._precision], reserved=0)
        for i in range(c.num_layers):
            cfg.depths[i], cfg.num_heads[i] = c.depths[i], c.num_heads[i]
        h = C.c_void_p(0)
        with torch.cuda.device(device):
            _lib.check(lib.ocm_swin_create(C.byref(cfg), C.byref(h)))
            for key, p in params:
                if p.device != device:
                    lib.ocm_swin_destroy(h)
                    raise RuntimeError(f"parameters are on {p.device} but the input is on {device}; call model.to(device)")
                t = p.detach().to(torch.float32).contiguous()
                _lib.check(lib.ocm_swin_set_param(h, key.encode(), _p(t), t.numel(), _stream()))
            torch.cuda.current_stream().synchronize()
        eng = dict(h=h, sig=sig, ws={})
        self.__dict__["_engine"] = eng
        return eng

    @torch.no_grad()
    def forward(self, pixel_values=None, labels=None, output_hidden_states=False, **unused):
        _require_hip(pixel_values, "pixel_values")
        c = self.config
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != (c.num_channels, c.image_size, c.image_size):
            raise ValueError(f"expected (B, {c.num_channels}, {c.image_size}, {c.image_size}) pixel_values, got "
                             f"{tuple(pixel_values.shape)}")
        x = pixel_values.detach().to(torch.float32).contiguous()
        dev, B = x.device, x.shape[0]
        eng, lib = self._get_engine(dev), _lib.load()
        side = c.image_size // c.patch_size
        for _ in range(c.num_layers - 1):  # SwinPatchMerging.maybe_pad: an odd grid is padded to an even one before it is halved
            side = (side + 1) // 2
        L = side * side
        logits = torch.empty((B, c.num_labels), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, c.hidden_size), dtype=torch.float32, device=dev)
        hidden = torch.empty((B, L, c.hidden_size), dtype=torch.float32, device=dev) if output_hidden_states else None
        ws = eng["ws"].get(B)
        if ws is None:
            nbytes = lib.ocm_swin_workspace_bytes(eng["h"], B)
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            eng["ws"] = {B: ws}
        off = (-ws.data_ptr()) % 256
        with torch.cuda.device(dev):
            _lib.check(lib.ocm_swin_forward(eng["h"], _p(x), B, _p(logits), _p(pooled), _p(hidden) if hidden is not None else None,
                                            C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream()))
        loss = None
        if labels is not None:
            loss = nn.functional.cross_entropy(logits, labels.to(dev))
        return types.SimpleNamespace(loss=loss, logits=logits, pooler_output=pooled, last_hidden_state=hidden)
