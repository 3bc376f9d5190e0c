"""Mirror of the reference's model.py wrappers around the ViT trunk:

  VisionTransformerForSimMIM    model.py:10-53    patch embed -> mask-token blend -> trunk -> (B,C,H,W)
  MIM                           model.py:55-83    encoder + Conv2d(1x1) + PixelShuffle decoder, masked L1 loss
  VisionTransformerForFinetune  model.py:110-139  trunk -> (B,C,H,W)
  LinearProbing                 model.py:142-174  encoder + one-layer (1x1 conv + PixelShuffle) decoder
  build_model / build_finetune_model / get_state_dict   model.py:85-108,176-226

Same constructor arguments, attributes and state_dict keys. The encoders run in one engine call (the mask
blend is fused into the patch-embedding epilogue, the (B,C,H,W) permute is a device transpose); the 1x1-conv
decoders run token-major as one MFMA GEMM + a pixel-shuffle kernel. Inference only: parameters are read,
never differentiated. The reference initialises mask_token with timm's trunc_normal_; here the package's
own trunc_normal_ (dino/utils.py) with the same bounds is used.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from . import _lib
from .dino.utils import trunc_normal_
from .dino.vision_transformer import VisionTransformer
from .engine import _p, _require_hip, _stream, to_operand


class _FmapEncoder(VisionTransformer):
    """Shared body of the two encoders: prepare tokens (optionally masked), all blocks, final norm, drop the
    CLS token and return the (B, C, H, W) map."""

    def _encode(self, x, mask=None, tokens=False):
        x = self._check_input(x)
        eng = self._engine(x.device)
        npatch = (x.shape[-2] // eng.p) * (x.shape[-1] // eng.p)
        side = self.img_size[0]
        if side != 224:  # model.py:38-39,124-125: positions interpolated for the CONFIGURED size
            pos = self._pos_for(npatch, side, side, x.device)
        else:  # model.py:40-41,126-127: x + self.pos_embed needs the native token count
            n0 = self.pos_embed.shape[1] - 1
            if npatch != n0:
                raise RuntimeError(f"The size of tensor a ({npatch + 1}) must match the size of tensor b ({n0 + 1}) "
                                   "at non-singleton dimension 1")
            pos = self._pos_for(n0, 224, 224, x.device)
        flags = _lib.OCM_OUT_FEAT if tokens else _lib.OCM_OUT_FMAP
        out = eng.forward(x, pos, flags=flags, patch_mask=mask)
        if tokens:
            return out["feat"][0]
        fmap = out["fmap"]
        B, Cc, hp, wp = fmap.shape
        side_t = int((hp * wp) ** 0.5)  # model.py:50-52: H = W = int(L ** 0.5)
        return fmap.reshape(B, Cc, side_t, side_t)


class VisionTransformerForSimMIM(_FmapEncoder):
    def __init__(self, interpolate_encoding=False, img_size=224, **kwargs):
        super().__init__(**kwargs)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        self.img_size = img_size
        self._trunc_normal_(self.mask_token, std=.02)
        self.interpolate_encoding = interpolate_encoding

    def _trunc_normal_(self, tensor, mean=0., std=1.):
        trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)

    def forward(self, x, mask):
        assert mask is not None
        return self._encode(x, mask=mask.to(x.device))


class VisionTransformerForFinetune(_FmapEncoder):
    def __init__(self, interpolate_encoding=False, img_size=224, **kwargs):
        super().__init__(**kwargs)
        self.img_size = img_size
        self.interpolate_encoding = interpolate_encoding

    def _trunc_normal_(self, tensor, mean=0., std=1.):
        trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)

    def forward(self, x):
        return self._encode(x)


def _conv1x1_pixel_shuffle(encoder, tokens, conv, stride, cache):
    """Conv2d(D, s*s*c, 1) + PixelShuffle(s) evaluated on the token-major normed tokens (B, N, D): one GEMM over
    the patch rows and a scatter. Returns (B, c, hp*s, wp*s) fp32."""
    B, N, D = tokens.shape
    O = conv.out_channels
    c_out = O // (stride * stride)
    hp = wp = int((N - 1) ** 0.5)
    prec = _lib.PRECISIONS[encoder._precision]
    dev = tokens.device
    lib = _lib.load()
    key = (conv.weight.data_ptr(), conv.weight._version, prec)
    if cache.get("key") != key:  # operand copy of the (O, D, 1, 1) weight in the engine's element type
        w32 = conv.weight.detach().reshape(O, D).to(device=dev, dtype=torch.float32).contiguous()
        w = to_operand(w32, prec)
        bias = (conv.bias.detach() if conv.bias is not None else torch.zeros(O)).to(device=dev, dtype=torch.float32)
        cache.update(key=key, w=w, bias=bias.contiguous())
    patches = tokens[:, 1:].contiguous()  # (B, P, D) fp32: drop the CLS row
    M = B * (N - 1)
    lin = torch.empty((M, O), dtype=torch.float32, device=dev)
    out = torch.empty((B, c_out, hp * stride, wp * stride), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        a = to_operand(patches.reshape(M, D), prec)
        _lib.check(lib.ocm_op_linear(prec, _p(a), _p(cache["w"]), _p(cache["bias"]), None, _p(lin), M, O, D,
                                     _lib.OCM_EPI_BIAS_F32, _stream()))
        _lib.check(lib.ocm_op_pixel_shuffle(_p(lin), _p(out), B, hp, wp, c_out, stride, _stream()))
    return out


def _pixel_shuffle_head(channels, stride, out_mult):
    """Conv2d(channels, stride^2 * out_mult, 1) -> PixelShuffle(stride): the decoder head of model.py:60-66,147-152.
    Kept as torch modules for their parameters / state_dict keys; the arithmetic runs in _conv1x1_pixel_shuffle."""
    return nn.Sequential(nn.Conv2d(channels, stride * stride * out_mult, kernel_size=1), nn.PixelShuffle(stride))


def _prefixed(encoder, method):
    fn = getattr(encoder, method, None)
    return {"encoder." + name for name in fn()} if fn is not None else {}


class MIM(nn.Module):
    """model.py:55-83: masked-image-modelling wrapper, forward(x, mask) -> (loss, x_rec, mask)."""

    def __init__(self, encoder, encoder_stride):
        super().__init__()
        self.encoder, self.encoder_stride = encoder, encoder_stride
        self.decoder = _pixel_shuffle_head(encoder.num_features, encoder_stride, 3)
        self.in_chans, self.patch_size = 3, 8
        self.__dict__["_dec_cache"] = {}

    @torch.no_grad()
    def forward(self, x, mask):
        _require_hip(x, "input")
        mask = mask.to(x.device)
        tokens = self.encoder._encode(x, mask=mask, tokens=True)
        x_rec = _conv1x1_pixel_shuffle(self.encoder, tokens, self.decoder[0], self.encoder_stride, self._dec_cache)
        # masked L1 reconstruction loss (model.py:71-73) — training bookkeeping, a handful of elementwise torch ops
        p = self.patch_size
        m32 = mask.to(torch.float32).contiguous()
        pixel_mask = torch.empty((m32.shape[0], m32.shape[1] * p, m32.shape[2] * p), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().ocm_op_nearest_upsample(_p(m32), _p(pixel_mask), m32.shape[0], m32.shape[1], m32.shape[2], p,
                                                           _stream()))
        pixel_mask = pixel_mask.to(mask.dtype).unsqueeze(1)  # 0 / 1 values: exact in the caller's dtype
        err = (x - x_rec).abs() * pixel_mask
        loss = err.sum() / (pixel_mask.sum() + 1e-5) / self.in_chans
        return loss, x_rec, pixel_mask

    @torch.jit.ignore
    def no_weight_decay(self):
        return _prefixed(self.encoder, "no_weight_decay")

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return _prefixed(self.encoder, "no_weight_decay_keywords")


class LinearProbing(nn.Module):
    """model.py:142-174: encoder + segmentation decoder (layer_num 1: 1x1 conv + PixelShuffle; 2: two 3x3 convs)."""

    def __init__(self, encoder, encoder_stride, layer_num=1):
        super().__init__()
        self.encoder, self.layer_num, self.encoder_stride = encoder, layer_num, encoder_stride
        feats, s2 = encoder.num_features, encoder_stride ** 2
        self.one_layer_decoder = _pixel_shuffle_head(feats, encoder_stride, 1)
        self.two_layer_decoder = nn.Sequential(
            nn.Conv2d(feats, 4 * s2, kernel_size=3, padding=1), nn.BatchNorm2d(4 * s2), nn.ReLU(inplace=True),
            nn.Conv2d(4 * s2, s2, kernel_size=3, padding=1), nn.PixelShuffle(encoder_stride))
        self.__dict__["_dec_cache"] = {}

    def _two_layer(self, tokens):
        """two_layer_decoder (model.py:154-166) in eval mode on the HIP path, token-major: im2col(3x3) + MFMA GEMM with
        the BatchNorm (running statistics) folded into the first convolution's weights, ReLU applied while gathering the
        second convolution's operand, PixelShuffle by the scatter kernel."""
        conv1, bn, _, conv2, _ = self.two_layer_decoder
        if bn.training:
            raise NotImplementedError("the HIP decoder evaluates BatchNorm with its running statistics: call .eval()")
        enc, dev, lib = self.encoder, tokens.device, _lib.load()
        prec = _lib.PRECISIONS[enc._precision]
        B, N, D = tokens.shape
        hp = wp = int((N - 1) ** 0.5)
        key = tuple((t.data_ptr(), t._version) for t in (conv1.weight, conv1.bias, bn.weight, bn.bias, bn.running_mean,
                                                         bn.running_var, conv2.weight, conv2.bias)) + (prec,)
        c = self._dec_cache
        if c.get("key2") != key:
            f32 = dict(device=dev, dtype=torch.float32)
            g = (bn.weight.detach().to(**f32) / torch.sqrt(bn.running_var.detach().to(**f32) + bn.eps))
            w1 = conv1.weight.detach().to(**f32) * g[:, None, None, None]
            b1 = (conv1.bias.detach().to(**f32) - bn.running_mean.detach().to(**f32)) * g + bn.bias.detach().to(**f32)
            # (O, C, 3, 3) -> (O, 3, 3, C): the K order of ocm_op_im2col3x3
            w1 = w1.permute(0, 2, 3, 1).reshape(w1.shape[0], -1).contiguous()
            w2 = conv2.weight.detach().to(**f32).permute(0, 2, 3, 1).reshape(conv2.out_channels, -1).contiguous()
            c.update(key2=key, w1=to_operand(w1, prec), b1=b1.contiguous(), w2=to_operand(w2, prec),
                     b2=conv2.bias.detach().to(**f32).contiguous())
        M, mid, out_c = B * hp * wp, conv1.out_channels, conv2.out_channels
        esz_t = {_lib.OCM_PREC_BF16: torch.bfloat16, _lib.OCM_PREC_FP32: torch.float32, _lib.OCM_PREC_BF16X3: torch.int32}[prec]
        patches = tokens[:, 1:].contiguous()  # (B, hp*wp, D) fp32
        out = torch.empty((B, 1, hp * self.encoder_stride, wp * self.encoder_stride), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            a1 = torch.empty((M, 9 * D), dtype=esz_t, device=dev)
            _lib.check(lib.ocm_op_im2col3x3(prec, _p(patches), _p(a1), B, hp, wp, D, 0, _stream()))
            y1 = torch.empty((M, mid), dtype=torch.float32, device=dev)
            _lib.check(lib.ocm_op_linear(prec, _p(a1), _p(c["w1"]), _p(c["b1"]), None, _p(y1), M, mid, 9 * D,
                                         _lib.OCM_EPI_BIAS_F32, _stream()))
            a2 = torch.empty((M, 9 * mid), dtype=esz_t, device=dev)
            _lib.check(lib.ocm_op_im2col3x3(prec, _p(y1), _p(a2), B, hp, wp, mid, 1, _stream()))  # ReLU on the way
            y2 = torch.empty((M, out_c), dtype=torch.float32, device=dev)
            _lib.check(lib.ocm_op_linear(prec, _p(a2), _p(c["w2"]), _p(c["b2"]), None, _p(y2), M, out_c, 9 * mid,
                                         _lib.OCM_EPI_BIAS_F32, _stream()))
            _lib.check(lib.ocm_op_pixel_shuffle(_p(y2), _p(out), B, hp, wp, out_c // self.encoder_stride ** 2,
                                                self.encoder_stride, _stream()))
        return out

    @torch.no_grad()
    def forward(self, x):
        _require_hip(x, "input")
        tokens = self.encoder._encode(x, tokens=True)
        if self.layer_num == 2:
            return self._two_layer(tokens)
        return _conv1x1_pixel_shuffle(self.encoder, tokens, self.one_layer_decoder[0], self.encoder_stride,
                                      self._dec_cache)


def build_model(args):
    """model.py:85-103: the MIM pre-training encoder — depth 4, THREE heads of 128 channels. Heads that are not 64 wide
    run the engine's generic fp32 attention kernel (kernels_attn.hip::attn_generic_kernel); everything else is the
    same MFMA path as the DINO-shaped encoders."""
    return VisionTransformerForSimMIM(patch_size=args.MODEL.PATCH_SIZE, embed_dim=384, depth=4, num_heads=3,
                                      mlp_ratio=4, img_size=[args.DATA.IMG_SIZE], qkv_bias=True,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6), interpolate_encoding=True)


def build_finetune_model(args):
    encoder = VisionTransformerForFinetune(patch_size=args.MODEL.PATCH_SIZE, embed_dim=384, depth=12, num_heads=6,
                                           mlp_ratio=4, img_size=[args.DATA.IMG_SIZE], qkv_bias=True,
                                           norm_layer=partial(nn.LayerNorm, eps=1e-6), interpolate_encoding=True)
    state_dict = get_state_dict(args)
    encoder.load_state_dict(state_dict, strict=False)
    return encoder


def get_state_dict(args):
    """model.py:190-226: a local checkpoint file is loaded (weights_only) and its `module.` / `backbone.`
    prefixes stripped. The reference's fallback downloads DINO weights from dl.fbaipublicfiles.com; there is
    no network on this path, so a missing file is an error."""
    if os.path.isfile(args.PRETRAINED_WEIGHTS):
        state_dict = torch.load(args.PRETRAINED_WEIGHTS, map_location="cpu", weights_only=True)
        key = getattr(args, "checkpoint_key", None)
        if key is not None and key in state_dict:
            state_dict = state_dict[key]
        state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
        state_dict = {k.replace("backbone.", ""): v for k, v in state_dict.items()}
        return state_dict
    raise FileNotFoundError(f"pretrained weights {args.PRETRAINED_WEIGHTS!r} not found (the reference would download "
                            "DINO weights here; no network on this path)")
