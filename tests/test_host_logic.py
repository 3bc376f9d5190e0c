"""CPU-side checks: the C-ABI library loads and exports every symbol include/ocm_vit.h declares,
argument validation / error translation, integer index math against the oracle's restatement,
and the nn.Module surface (state_dict keys, loud failure without a HIP device). No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import vit_oracle as O
from tests.helpers import CASES, build_module, case_state_dict
from vit_ocm_wmsegmentation_amd import _lib, sw_processing, synth
from vit_ocm_wmsegmentation_amd import utils as amd_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "ocm_vit.h")).read() + open(os.path.join(ROOT, "include", "ocm_swin.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ocm_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"libocm_vit.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.ocm_abi_version() == _lib.OCM_ABI_VERSION
    # ... and nothing BUT the C ABI: no mangled launcher, kernel host stub or global leaks into the dynamic symbol table
    # (csrc/exports.map; development-only ocm_debug_* entry points exist in `make dev` builds, never in the shipped library)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported, "nm found no dynamic symbols"
    stray = sorted(n for n in exported if not n.startswith("ocm_"))
    assert not stray, f"non-ABI symbols exported: {stray[:5]} (+{max(0, len(stray) - 5)})"
    assert not [n for n in exported if n.startswith("ocm_debug")], "development entry points in the shipped library"
    assert declared <= exported


def test_create_rejects_bad_configs(lib):
    def create(**kw):
        base = dict(patch_size=16, in_chans=3, embed_dim=384, depth=12, num_heads=6, mlp_hidden=1536, ln_eps=1e-6,
                    qk_scale=0.125, precision=0, reserved=0)
        base.update(kw)
        cfg = _lib.OcmVitConfig(**base)
        h = C.c_void_p(0)
        return lib.ocm_vit_create(C.byref(cfg), C.byref(h))

    for bad in (dict(embed_dim=100), dict(num_heads=5), dict(patch_size=7), dict(in_chans=2), dict(depth=0),
                dict(mlp_hidden=100), dict(precision=7), dict(precision=3)):
        assert create(**bad) == _lib.OCM_EINVAL, bad
        assert lib.ocm_last_error()
    with pytest.raises(ValueError):
        _lib.check(create(embed_dim=100))


def test_ops_validate_arguments(lib):
    assert lib.ocm_op_linear(0, None, None, None, None, None, 1, 32, 64, 0, None) == _lib.OCM_EINVAL
    one = C.c_void_p(256)
    assert lib.ocm_op_linear(0, one, one, None, None, one, 10, 33, 64, 0, None) == _lib.OCM_EINVAL  # N % 32
    assert lib.ocm_op_linear(0, one, one, None, None, one, 10, 32, 60, 0, None) == _lib.OCM_EINVAL  # K % 64
    assert lib.ocm_op_linear(0, one, one, None, None, one, 10, 32, 64, 1, None) == _lib.OCM_EINVAL  # resid missing
    assert lib.ocm_op_linear(7, one, one, None, None, one, 10, 32, 64, 0, None) == _lib.OCM_EINVAL  # precision
    assert lib.ocm_op_attention(0, one, one, one, None, None, 1, 10, 1, 0.125, None) == _lib.OCM_EINVAL
    assert lib.ocm_op_attention_map(one, one, 0, 3, 10, 0, 2, 2, 8, None) == _lib.OCM_EINVAL  # hf*wf+1 != N


@pytest.mark.parametrize("size,stride", [(1152, 128), (4096, 128), (384, 128), (256, 128), (1000, 100), (130, 64)])
def test_sliding_window_index_math_bit_exact(size, stride):
    ref = O.sliding_window_origins(size, size, stride)
    got = sw_processing.sliding_window_origins(size, size, stride)
    assert got.dtype == np.int32 and got.shape == (len(ref), 2)
    assert [tuple(r) for r in got.tolist()] == ref
    assert sw_processing.window_count(size, stride) == len(range(0, size - 2 * stride, stride))


def test_sliding_window_origins_match_reference_golden():
    from tests.helpers import load_golden
    gold = load_golden("helpers")  # origins produced by the reference's own sliding_window
    for size in (640, 1152):
        assert np.array_equal(sw_processing.sliding_window_origins(size, size, 128), gold[f"sw_origins_{size}"])


def test_out_of_bounds_windows_are_zero_filled_like_pil_crop():
    """sw_processing.py:157-160 crops with PIL; a slab whose side is not a multiple of the stride (or window > 3*stride)
    has windows that reach past the image, which PIL zero-fills. The oracle's crops must equal PIL's."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for size, stride, window in ((200, 32, 96), (160, 32, 128)):  # 200: not a multiple; 160/128: window = 4 * stride
        arr = rng.integers(1, 255, (size, size), dtype=np.uint8)
        img = Image.fromarray(arr)
        origins = O.sliding_window_origins(size, size, stride)
        assert max(y for y, _ in origins) + window > size  # the case under test
        pil = np.stack([np.asarray(img.crop((x, y, x + window, y + window))) for y, x in origins])
        ours = O.sliding_window_crops(torch.from_numpy(arr.astype(np.float32))[None], stride, window)[:, 0].numpy()
        assert np.array_equal(pil.astype(np.float32), ours)


def test_sliding_window_rectangular():
    ref = O.sliding_window_origins(768, 1152, 128)
    got = sw_processing.sliding_window_origins(768, 1152, 128)
    assert [tuple(r) for r in got.tolist()] == ref and len(ref) == 4 * 7


@pytest.mark.parametrize("n,world", [(900, 8), (49, 8), (49, 2), (7, 8), (0, 4), (16, 4), (1, 1)])
def test_shard_partition(n, world):
    share = -(-n // world) if n else 0
    covered = []
    for r in range(world):
        b, e, s = sw_processing.shard_range(n, world, r)
        assert s == share and 0 <= b <= e <= n and e - b <= share
        assert b == min(r * share, n)
        covered += list(range(b, e))
    assert covered == list(range(n))  # contiguous, complete, in order
    if (n, world) == (900, 8):
        assert [sw_processing.shard_range(n, world, r)[1] - sw_processing.shard_range(n, world, r)[0]
                for r in range(8)] == [113] * 7 + [109]


def test_region_query_index_bit_exact():
    for py, px, p, wf in [(0, 0, 16, 14), (223, 223, 16, 14), (37, 90, 16, 14), (383, 5, 8, 48), (100, 200, 8, 48)]:
        assert amd_utils.region_query_index(py, px, p, wf) == O.region_query_index(py, px, p, wf)
    assert amd_utils.grid_query_index(3, 2, 14, 2) == 3 * 14 * 2 + 2 * 2


def test_state_dict_keys_match_reference_layout():
    case = CASES["tiny_p8"]
    model = build_module(case, "cpu")
    sd = model.state_dict()
    shapes = synth.param_shapes(128, 2, 8, img_size=32)
    assert list(sd.keys()) == list(shapes.keys())  # same keys, same order as the reference module
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == shp
    ref = case_state_dict(case)
    for k in ref:
        assert torch.equal(sd[k], ref[k])
    # surface used by eval.py / model.py
    for attr in ("patch_embed", "cls_token", "pos_embed", "pos_drop", "blocks", "norm", "head", "embed_dim",
                 "num_features", "forward", "forward_feats", "get_intermediate_feat", "get_last_selfattention",
                 "get_intermediate_layers", "prepare_tokens", "interpolate_pos_encoding"):
        assert hasattr(model, attr)
    assert model.patch_embed.patch_size == 8 and model.patch_embed.num_patches == 16 and len(model.blocks) == 2


def test_factories_and_strict_false_loading():
    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    m = vits.__dict__["vit_small"](patch_size=16, num_classes=0)  # eval.py:60
    assert m.embed_dim == 384 and len(m.blocks) == 12 and m.blocks[0].attn.num_heads == 6
    assert sum(p.numel() for p in m.parameters()) == 21665664
    sd = {"module." + k: v for k, v in synth.synth_arch_state_dict("vit_small", 16, variant="init").items()}
    sd = {k.replace("module.", ""): v for k, v in sd.items()}  # eval.py:73-76
    sd.pop("blocks.11.mlp.fc2.bias")
    msg = m.load_state_dict(sd, strict=False)
    assert msg.missing_keys == ["blocks.11.mlp.fc2.bias"]
    assert vits.vit_tiny().embed_dim == 192 and vits.vit_base(patch_size=8).patch_embed.patch_size == 8


def test_interpolate_pos_encoding_matches_oracle():
    case = CASES["tiny_p8"]
    model = build_module(case, "cpu")
    sd = case_state_dict(case)
    for (w, h) in [(32, 32), (48, 32), (64, 64), (40, 56)]:
        npatch = (w // 8) * (h // 8)
        tok = torch.zeros(1, npatch + 1, 128)
        got = model.interpolate_pos_encoding(tok, w, h)
        ref = O.interpolate_pos_encoding(sd, npatch, w, h, 8)
        assert got.shape == ref.shape and torch.equal(got, ref)


def test_no_cpu_fallback_fails_loudly():
    """The product path must not silently run on the CPU (or through the oracle)."""
    model = build_module(CASES["tiny_p8"], "cpu")
    x = synth.synth_tiles(1, 32)
    for call in (model, model.get_last_selfattention, model.forward_feats, model.prepare_tokens,
                 lambda t: model.get_intermediate_feat(t, 1), lambda t: model.blocks[0](torch.zeros(1, 17, 128)),
                 lambda t: model.norm(torch.zeros(1, 17, 128))):
        with pytest.raises(RuntimeError, match="HIP"):
            call(x)
    with pytest.raises(RuntimeError, match="HIP"):
        amd_utils.compute_attention([torch.zeros(1, 2, 17, 17)], 0, 4, 4, 8)


def test_trunc_normal_statistics():
    from vit_ocm_wmsegmentation_amd.dino.utils import trunc_normal_
    torch.manual_seed(0)
    t = trunc_normal_(torch.empty(200000), std=.02)
    assert abs(t.mean().item()) < 2e-4 and abs(t.std().item() - 0.02) < 2e-4 and t.abs().max().item() < 0.12
    u = trunc_normal_(torch.empty(20000), mean=0., std=1., a=-1., b=1.)
    assert u.min().item() >= -1 and u.max().item() <= 1


def test_missing_extension_fails_loudly(tmp_path):
    """Without libocm_vit.so the compute path raises at load time — there is no Python / torch fallback to hide
    behind (checked in a fresh interpreter with OCM_VIT_LIB pointing at a file that does not exist)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from vit_ocm_wmsegmentation_amd import _lib\n"
            "try:\n"
            "    _lib.load()\n"
            "except _lib.OcmError as e:\n"
            "    assert 'not built' in str(e) and 'no CPU' in str(e).replace('There is no CPU', 'no CPU'), str(e)\n"
            "    print('LOUD')\n" % ROOT)
    env = dict(os.environ, OCM_VIT_LIB=str(tmp_path / "absent.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LOUD" in out.stdout, out.stdout + out.stderr


def test_engine_parameter_slots_follow_replaced_parameters():
    """The module resolves (owning module, key) pairs of its parameters once (a named_parameters() walk per one-tile call cost
    more than the call's launches); a Parameter object replaced afterwards is still what the next signature sees."""
    import torch.nn as nn
    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    m = vits.vit_tiny(patch_size=16, num_classes=0)
    named = dict(m._named_engine_params())
    want = {n for n, _ in m.named_parameters() if not n.startswith(("pos_embed", "head."))}
    assert set(named) == want
    old = m.blocks[1].attn.qkv.weight
    m.blocks[1].attn.qkv.weight = nn.Parameter(torch.zeros_like(old))
    again = dict(m._named_engine_params())
    assert again["blocks.1.attn.qkv.weight"] is m.blocks[1].attn.qkv.weight and again["blocks.1.attn.qkv.weight"] is not old
    with torch.no_grad():
        m.blocks[0].mlp.fc1.bias.add_(1.0)  # in-place update: same object, new version -> a different signature
    sig0 = tuple((p.data_ptr(), p._version) for _, p in m._named_engine_params())
    with torch.no_grad():
        m.blocks[0].mlp.fc1.bias.add_(1.0)
    assert sig0 != tuple((p.data_ptr(), p._version) for _, p in m._named_engine_params())
    import pickle
    m2 = pickle.loads(pickle.dumps(m))  # the cached slots hold module references: they are not part of the pickled state
    assert set(dict(m2._named_engine_params())) == want
    # a replaced SUB-MODULE is followed too (the cache used to keep reading the old module's parameters)
    new_qkv = nn.Linear(192, 576)
    m.blocks[2].attn.qkv = new_qkv
    again = dict(m._named_engine_params())
    assert again["blocks.2.attn.qkv.weight"] is new_qkv.weight and again["blocks.2.attn.qkv.bias"] is new_qkv.bias
    blk = vits.Block(192, 3, qkv_bias=True)
    m.blocks[3] = blk
    assert dict(m._named_engine_params())["blocks.3.mlp.fc2.weight"] is blk.mlp.fc2.weight
    m.blocks[0].attn.proj.bias = None  # a parameter removed after caching drops out instead of raising
    assert "blocks.0.attn.proj.bias" not in dict(m._named_engine_params())
    assert set(dict(m._named_engine_params())) == want - {"blocks.0.attn.proj.bias"}


def test_stress_variant_is_calibrated_per_geometry():
    assert synth.stress_variant("vit_small", 16) == "peaked" and synth.qkv_gain_of("peaked") == 8.0
    assert synth.qkv_gain_of(synth.stress_variant("vit_base", 16)) == 6.5
    assert synth.qkv_gain_of(synth.stress_variant("vit_small", 8)) == 10.0
    with pytest.raises(ValueError):
        synth.qkv_gain_of("qkvx")
