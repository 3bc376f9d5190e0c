"""Rebuild one precision's section of profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries that
tools/pmc_summary.py prints (values in KiB per launch).  traffic = (2*FETCH + WRITE) * 1024: FETCH_SIZE is doubled per
MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE is exact for 16-B/lane stores.

    python tools/pmc_traffic.py profiles/r02_pmc_fetch.txt profiles/r02_pmc_write.txt bf16x3 [profiles/r02_pmc_tcc.txt]
"""
import json
import os
import re
import sys

# kernel class -> leading text of the kernel name (after "void "), per precision mode
CLASSES = {
    "bf16x3": {
        # round 4: GemmCfg carries the MFMA shape as its fifth argument (1 = v_mfma_f32_16x16x32_bf16: fc1 / qkv)
        "fc1_gemm": "gemm_dma_kernel<GemmCfg<160, 128, 2, 4, 1>, sp32, false, 12,",  # 160-row tiles (GemmCfg::HALF) at this row count
        # round 3: LayerNorm folded into its consumer -> attn.proj / mlp.fc2 on 128 x 192 LDS-DMA tiles (x, split(x), row sums)
        "fc2_gemm": "gemm_dma_kernel<GemmCfg<128, 192, 4, 2, 0>, sp32, false, 48,",
        "proj_gemm": "gemm_dma_kernel<GemmCfg<128, 192, 4, 2, 0>, sp32, false, 12,",
        "qkv_gemm": "qkv_dma_kernel<GemmCfg<128, 128, 2, 4, 1>, sp32, 12, 2>",
        "attention": "attn_fwd_x3_pp_kernel<true>",  # round 4: the software-pipelined kernel up to 1 024 tokens
        "patch_embed": "gemm_kernel<GemmCfg<64, 128, 2, 2, 0>, sp32, false, 24, PatchLoader<",
    },
    # config 4 (ViT-S/8 slab sweep, 20-21 windows of 2305 tokens per launch): summaries of tools/sweep_slab.py runs
    "slab_bf16x3": {
        "attention": "attn_fwd_x3_dma_kernel<true, 8, 2,",
        "fc2_gemm": "gemm_dma_kernel<GemmCfg<128, 192, 4, 2, 0>, sp32, false, 48, 3, EpiLinear<1",  # 128 x 192 tiles, LayerNorm kernels at this size
        "fc1_gemm": "gemm_dma_kernel<GemmCfg<256, 256, 2, 4, 1>, sp32, false, 12, 2, EpiLinear<2",  # round 4: 256 x 256 tiles at K = 384 from 16 k rows
    },
}

D, HID = 384, 1536


def algorithmic_bytes(prec):
    e = 4 if prec.endswith("bf16x3") else 2                 # operand bytes per element
    # tokens per launch: the bench workload (ViT-S/16, 224^2, 64 tiles of 197) or the slab sweep's mean launch
    # (900 windows of 2305 tokens in 43 forwards of 20-21: SlidingWindowAttention.auto_batch_plan)
    T = round(900 / 43 * 2305) if prec.startswith("slab") else 64 * 197
    qkv_out = 3 * T * D * e                                 # q, k, v^T: valid tokens only (pad rows are never written)
    return {
        "fc1_gemm": T * D * e + D * HID * e + T * HID * e,
        # + residual in / out (fp32); the bench workload's fused kernel also writes xn
        "fc2_gemm": T * HID * e + HID * D * e + T * D * 4 * 2 + (0 if prec.startswith("slab") else T * D * e),
        "proj_gemm": T * D * e + D * D * e + T * D * 4 * 2 + T * D * e,
        "qkv_gemm": T * D * e + D * 3 * D * e + qkv_out,
        "attention": qkv_out + T * D * e,
        "patch_embed": 64 * 3 * 224 * 224 * 4 + 768 * D * e + T * D * 4 + T * D * e,  # + split(x) for the folded first LayerNorm
    }


def _core(name):
    return name.strip().removeprefix("void ")


def parse(path):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.rstrip("\n")
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
            if m and name is not None:
                out.setdefault(name, {})[m.group(1)] = float(m.group(2))
    return out


def main():
    fetch, write, prec = parse(sys.argv[1]), parse(sys.argv[2]), sys.argv[3]
    tcc = parse(sys.argv[4]) if len(sys.argv) > 4 else {}
    alg = algorithmic_bytes(prec)
    sec = {}
    for cls, must in CLASSES[prec].items():
        # pmc_summary.py truncates long names: a kernel matches when one text is a prefix of the other
        names = [n for n in fetch if _core(n).startswith(must) or must.startswith(_core(n))]
        if len(names) != 1:
            raise SystemExit(f"{cls}: {len(names)} kernels match {must!r}")
        n = names[0]
        f, w = fetch[n]["FETCH_SIZE"], write[n]["WRITE_SIZE"]
        sec[cls] = {"fetch_kib": f, "write_kib": w, "traffic_bytes": int((2 * f + w) * 1024),
                    "algorithmic_bytes": int(alg[cls])}
        if n in tcc:
            h, m = tcc[n]["TCC_HIT_sum"], tcc[n]["TCC_MISS_sum"]
            sec[cls]["l2_hit_rate"] = round(h / (h + m), 3)
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    d = json.load(open(p))
    d[prec] = sec
    json.dump(d, open(p, "w"), indent=1)
    for k, v in sec.items():
        print(f"{k:12s} traffic {v['traffic_bytes'] / 1e6:8.1f} MB  algorithmic {v['algorithmic_bytes'] / 1e6:8.1f} MB"
              f"  ratio {v['traffic_bytes'] / v['algorithmic_bytes']:.2f}")


if __name__ == "__main__":
    main()
