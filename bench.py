#!/usr/bin/env python3
"""bench.py — OCM tiles/s of the ViT attention-map hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ViT-S/16, batch 64 of 224x224 synthetic OCM tiles per GPU,
bf16 MFMA operands / fp32 accumulate, attention-map extraction = `get_last_selfattention`
(full (B,H,N,N) fp32 probabilities of the last block) + the CLS-row maps callers consume.
A step = one such forward over one resident batch on every rank (weak scaling: the tiles of a
sweep are independent) followed by the all-gather of the per-tile CLS-row maps
(RCCL, overlapped with the next step's compute). Inputs and weights are in HBM before the timed
region. Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_FP32_MFMA_TFLOPS = 157.3    # same guide: f32-input MFMA = 1/16 of the bf16 rate


def flops_per_tile(D, L, p, S, in_chans=3):
    """Algorithmic MFMA FLOPs (1 MAC = 2), SURVEY §8-d: F_map = F_pe + (L-1) F_block + 6ND^2 + 2N^2D."""
    P = (S // p) ** 2
    N = P + 1
    f_block = 24 * N * D * D + 4 * N * N * D
    f_pe = 2 * P * in_chans * p * p * D
    return f_pe + (L - 1) * f_block + 6 * N * D * D + 2 * N * N * D, N


def class_flops(D, M, N, B, p, in_chans):
    """Algorithmic FLOPs of ONE launch of each kernel class at batch B (per-unit figure x units)."""
    T, P = B * N, N - 1
    return {
        "patch_embed": 2.0 * B * P * in_chans * p * p * D,
        "qkv_gemm": 2.0 * T * D * 3 * D,
        "attention": 4.0 * B * N * N * D,  # QK^T + PV over all heads
        "proj_gemm": 2.0 * T * D * D,
        "fc1_gemm": 2.0 * T * D * M,
        "fc2_gemm": 2.0 * T * M * D,
    }


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    exposes every core of the host but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if "OCM_CPU_THREADS" in os.environ:
        n = int(os.environ["OCM_CPU_THREADS"])
    return n


def cpu_baseline(arch_dims, patch, size, batch, seconds_budget=20.0):
    """The oracle (CPU restatement of the reference, fp32 torch-CPU ops) timed on this host: at the bench batch and at
    B = 1, which is how the reference's own loops call the model (eval.py:128-136, sw_processing.py:235-239)."""
    from oracle import vit_oracle as O  # CPU baseline leg: allowed importer of oracle/
    from vit_ocm_wmsegmentation_amd import synth
    D, L, H = arch_dims
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synth.synth_state_dict(D, L, patch, seed=0, variant="init")
    cfg = O.make_cfg(sd, patch, H)
    x = synth.synth_tiles(batch, size)
    O.get_last_selfattention(sd, cfg, x)  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while iters < 5 and (iters < 2 or time.perf_counter() - t0 < seconds_budget):
        O.get_last_selfattention(sd, cfg, x)
        iters += 1
    dt = time.perf_counter() - t0
    x1 = x[:1]
    O.get_last_selfattention(sd, cfg, x1)
    t1 = time.perf_counter()
    it1 = 0
    while it1 < 40 and time.perf_counter() - t1 < 4.0:
        O.get_last_selfattention(sd, cfg, x1)
        it1 += 1
    dt1 = time.perf_counter() - t1
    return {"value": round(batch * iters / dt, 2), "unit": "tiles/s", "cores": cores, "kind": "port",
            "b1_value": round(it1 / dt1, 2),
            "sample": f"{iters} batches of {batch} tiles ({size}x{size}, get_last_selfattention, fp32 torch-CPU "
                      f"oracle, {cores} threads) after 1 warm-up; b1_value = {it1} single-tile calls (the reference's "
                      f"callers run B = 1)"}


def spread(values):
    """[min, median, max] of the per-rank figures, 3 decimals."""
    v = sorted(values)
    return [round(v[0], 3), round(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3), round(v[-1], 3)]


def slab_sweep(args, dev, world, rank, lib):
    """BASELINE.json configs[3] — the path north_star prices at 8 GPUs: ViT-S/8 sliding-window sweep of a 4096^2 slab
    (900 windows of 384^2, N = 2305) sharded over the ranks with one all-gather of the CLS-row maps. STRONG scaling:
    the slab is fixed, every rank runs its block of windows. All ranks call this; rank 0 returns the report."""
    import ctypes as C

    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    from vit_ocm_wmsegmentation_amd import _lib, synth
    from vit_ocm_wmsegmentation_amd.sw_processing import SlidingWindowAttention
    model = vits.vit_small(patch_size=8, num_classes=0)
    model.load_state_dict(synth.synth_arch_state_dict("vit_small", 8, seed=0, variant="init"))
    model.eval().to(dev).set_precision(args.precision)
    slab = synth.synth_tiles(1, args.slab_size, seed=7)[0].to(dev)
    sweep = SlidingWindowAttention(model, window=384, stride=128, batch_tiles="auto")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    maps = sweep(slab)  # warm-up (engine build, workspace, RCCL channel)
    sync()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        maps = sweep(slab)
    sync()
    dt = (time.perf_counter() - t0) / reps
    # every rank's own figures travel to rank 0: its sweep time, its share of the windows and how many forwards its plan has
    # (an imbalance — 113 / 109 windows, auto plans of different length — must be visible next to the max-over-ranks time)
    from vit_ocm_wmsegmentation_amd.sw_processing import shard_range
    b_own, e_own, _ = shard_range(maps.shape[0], world, rank)
    own_plan = sweep._plan(e_own - b_own, 2305, dev)
    mine = torch.tensor([dt, e_own - b_own, len(own_plan)], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    per_rank = [[float(v) for v in t.tolist()] for t in per_rank]
    dt = max(r[0] for r in per_rank)
    # dominant kernel of a rank's share, timed with HIP events on the launch stream. The sweep ends in a collective, so
    # EVERY rank runs this extra pass (each profiles its own launches); rank 0 reports its figures.
    roof = None
    ms = (C.c_double * len(_lib.KERNEL_CLASSES))()
    cnt = (C.c_int64 * len(_lib.KERNEL_CLASSES))()
    _lib.check(lib.ocm_prof_begin(0xFFFFFFFF, 4096))
    sweep(slab)
    torch.cuda.synchronize()
    _lib.check(lib.ocm_prof_end(ms, cnt))
    if rank == 0:
        D, Hh, N = 384, 6, 2305
        b0, e0, plan = b_own, e_own, own_plan  # windows per forward on this rank (auto: whole rounds of the CUs)
        cf1 = class_flops(D, 4 * D, N, 1, 8, 3)
        cf = {c: v * (e0 - b0) / len(plan) for c, v in cf1.items()}  # mean FLOPs per launch
        dom = max((c for c in _lib.KERNEL_CLASSES if c in cf), key=lambda c: ms[_lib.KERNEL_CLASSES.index(c)])
        i = _lib.KERNEL_CLASSES.index(dom)
        avg_s = ms[i] / max(cnt[i], 1) * 1e-3
        mpp = 3 if args.precision == "bf16x3" else 1
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_DENSE_TFLOPS
        ach = cf[dom] / avg_s / 1e12
        try:    # HBM-side bytes per launch of that kernel from the slab's own PMC passes (tools/pmc_traffic.py)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["slab_" + args.precision][dom]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            traffic = None
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "launches": int(cnt[i]), "avg_launch_us": round(avg_s * 1e6, 2),
                "flop_per_launch": cf[dom], "mfma_per_product": mpp, "mfma_pipe_frac": round(ach * mpp / peak, 4),
                "note": f"per launch of {min(plan)}-{max(plan)} windows ({len(plan)} forwards per sweep on this rank)"}
    T = maps.shape[0]
    fwin, _ = flops_per_tile(384, 12, 8, 384)
    if rank != 0:
        return None
    return {"workload": f"vit_small patch 8, {args.slab_size}^2 slab -> {T} windows of 384^2 (N=2305), up to 24 windows per forward (auto), "
                        f"CLS-row maps, tile shard x{world} + one all-gather", "scaling": "strong", "n_gpus": world,
            "windows": T, "ms_per_sweep": round(dt * 1e3, 2), "value": round(T / dt, 1), "unit": "windows/s",
            "per_rank_ms": spread([r[0] * 1e3 for r in per_rank]), "windows_per_rank": [int(r[1]) for r in per_rank],
            "forwards_per_rank": [int(r[2]) for r in per_rank],
            "path_tflops": round(T / dt * fwin / 1e12, 2), "roofline": roof}


def self_launch(n):
    """Run this script as n ranks under torch.distributed.run on 127.0.0.1 (a free port) and return their exit status."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--arch", default="vit_small")
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "fp32"],
                    help="operand precision of the contraction kernels. Default: bf16x3 (split-bf16 on the bf16 MFMA), the "
                         "fastest mode that holds the north star's 1e-3 on every golden weight set with an unsaturated softmax (DESIGN.md 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-slab", action="store_true", help="skip the sharded 4096^2 slab sweep (BASELINE configs[3]) extra")
    ap.add_argument("--slab-size", type=int, default=4096)
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel-class table to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started directly: become the launcher. Nothing in this process has touched HIP yet
        # (importing torch does not), so the N ranks are fresh children of torch.distributed.run, one per GPU; their
        # stdout is ours (rank 0 prints the JSON line) and their exit status is ours.
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N does that itself)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device")
    # one rank per GPU (RCCL). OCM_BENCH_BACKEND=gloo lets the N>1 code path be rehearsed on a 1-GPU box
    # (ranks share device 0, the all-gather goes through gloo); it is never the measured configuration.
    backend = os.environ.get("OCM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} HIP device(s) visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    from vit_ocm_wmsegmentation_amd import _lib, synth
    lib = _lib.load()
    # CPU-side torch work in this process (synthetic weights, the parity probes) on a few threads only: a full-width OpenMP team
    # keeps spinning after every parallel region and competes with the thread that launches kernels. cpu_baseline() sets the
    # full core count for itself at the very end.
    torch.set_num_threads(max(1, min(4, host_cores() // max(1, world))))

    D, L, H = synth.ARCHS[args.arch]
    B, S, p = args.batch, args.size, args.patch
    model = vits.__dict__[args.arch](patch_size=p, num_classes=0)
    model.load_state_dict(synth.synth_arch_state_dict(args.arch, p, seed=0, variant="init"))
    for q in model.parameters():
        q.requires_grad = False
    model.eval().to(dev).set_precision(args.precision)
    x = synth.synth_tiles(B, S, seed=1234 + rank).to(dev)  # resident in HBM before timing
    fmap, N = flops_per_tile(D, L, p, S)

    flags = _lib.OCM_OUT_ATTN | _lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY
    gathered = torch.empty((world * B, H, 1, N - 1), dtype=torch.float32, device=dev) if world > 1 else None
    pending = None

    def step():
        nonlocal pending
        out = model._run(x, flags=flags)  # full (B,H,N,N) attention + CLS rows (B,H,1,N-1)
        if world > 1:
            if pending is not None:
                pending.wait()
            pending = dist.all_gather_into_tensor(gathered, out["rows"], async_op=True)
        return out

    def sync():
        nonlocal pending
        if pending is not None:
            pending.wait()
            pending = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()

    ms = (C.c_double * len(_lib.KERNEL_CLASSES))()
    cnt = (C.c_int64 * len(_lib.KERNEL_CLASSES))()
    cf = class_flops(D, int(D * 4), N, B, p, 3)

    # pick the dominant kernel class (largest share of device time) with an untimed all-class pass ...
    _lib.check(lib.ocm_prof_begin(0xFFFFFFFF, 8 * L + 16))
    model._run(x, flags=flags)
    torch.cuda.synchronize()
    _lib.check(lib.ocm_prof_end(ms, cnt))
    dom = max((c for c in _lib.KERNEL_CLASSES if c in cf), key=lambda c: ms[_lib.KERNEL_CLASSES.index(c)])
    # ... and measure THAT class live with HIP events on the launch stream inside the timed region
    dom_idx = _lib.KERNEL_CLASSES.index(dom)
    _lib.check(lib.ocm_prof_begin(1 << dom_idx, args.steps * L + 8))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    _lib.check(lib.ocm_prof_end(ms, cnt))
    dom_ms, dom_n = ms[dom_idx], cnt[dom_idx]

    tall = [torch.tensor([dt], dtype=torch.float64, device=dev) for _ in range(world)]
    if world > 1:
        dist.all_gather(tall, tall[rank].clone())
    rank_ms = [float(t.item()) / args.steps * 1e3 for t in tall]  # ms per step as each rank saw it
    dt = max(float(t.item()) for t in tall)                       # the contract: MAX over ranks

    # per-class breakdown (separate, untimed pass: events around every launch)
    breakdown = None
    if rank == 0:
        reps = 5
        _lib.check(lib.ocm_prof_begin(0xFFFFFFFF, reps * (8 * L + 8)))
        for _ in range(reps):
            model._run(x, flags=flags)
        torch.cuda.synchronize()
        _lib.check(lib.ocm_prof_end(ms, cnt))
        breakdown = {}
        for i, name in enumerate(_lib.KERNEL_CLASSES):
            if cnt[i]:
                avg_us = ms[i] / cnt[i] * 1e3
                breakdown[name] = {"launches_per_fwd": cnt[i] // reps, "avg_us": round(avg_us, 2),
                                   "us_per_fwd": round(ms[i] / reps * 1e3, 1)}
                if name in cf:
                    breakdown[name]["tflops"] = round(cf[name] / (avg_us * 1e-6) / 1e12, 1)
        if args.breakdown:
            tot = sum(v["us_per_fwd"] for v in breakdown.values())
            print(f"per-class device time per forward (sum {tot:.0f} us):", file=sys.stderr)
            for k, v in breakdown.items():
                print(f"  {k:12s} {v}", file=sys.stderr)

    # the sharded slab sweep (strong scaling; all ranks take part), outside the timed region of the headline metric
    default_workload = (args.arch, p, S) == ("vit_small", 16, 224)
    slab = None
    if default_workload and not args.no_slab:
        del out
        slab = slab_sweep(args, dev, world, rank, lib)
        out = model._run(x, flags=flags)

    # Everything collective is done. The other ranks leave the process group NOW, so that what follows on rank 0 — the parity
    # probe and the CPU baseline, which claims the host's cores — runs with nobody parked in an RCCL barrier beside it.
    comm = {"backend": backend, "visible_devices": ndev, "world": world}
    if world > 1:
        if backend == "nccl":
            try:
                comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:  # noqa: BLE001 - version query only
                comm["rccl_version"] = f"unavailable ({type(exc).__name__})"
        names = [None] * world
        dist.all_gather_object(names, f"{torch.cuda.get_device_name(dev)} #{torch.cuda.current_device()} pid {os.getpid()}")
        comm["ranks"] = names
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # the single-bf16 fast mode next to the headline (rank 0, same batch, no collective): what the extra precision costs.
    # Timed BEFORE the CPU oracle runs below: their OpenMP workers keep spinning for a while after each parallel region and
    # take the launching thread's core (the leg read 1.5 / 3.0 / 4.4 ms per step on different boxes when it ran after them,
    # with the same 35 us fc1 launches under the events)
    fast = ob_pick = None
    if args.precision == "bf16x3" and default_workload:
        model.load_state_dict(synth.synth_arch_state_dict(args.arch, p, seed=0, variant="peaked"))
        model.set_precision("bf16")
        ob_pick = model._run(x, flags=flags)["attn"][0][sorted({0, B // 2, B - 1})].cpu()
        for _ in range(3):
            model._run(x, flags=flags)
        torch.cuda.synchronize()
        _lib.check(lib.ocm_prof_begin(1 << _lib.KERNEL_CLASSES.index("fc1_gemm"), args.steps * L + 8))
        tb = time.perf_counter()
        for _ in range(args.steps):
            model._run(x, flags=flags)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        _lib.check(lib.ocm_prof_end(ms, cnt))
        i1 = _lib.KERNEL_CLASSES.index("fc1_gemm")
        fc1_s = ms[i1] / max(cnt[i1], 1) * 1e-3
        fast = {"dtype": "bf16", "value": round(B * args.steps / dtb, 1), "unit": "tiles/s (this rank)",
                "ms_per_step": round(dtb / args.steps * 1e3, 4), "attn_linf_peaked_weights": None,
                "note": "single bf16 MFMA operands: ~1.75x the throughput, but 4-8e-2 off the reference on peaked (trained-like) "
                        "attention, 40-80x the north star's 1e-3 -> not the headline",
                "roofline": {"bound": "mfma", "kernel": "fc1_gemm", "achieved": round(cf["fc1_gemm"] / fc1_s / 1e12, 2),
                             "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(cf["fc1_gemm"] / fc1_s / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                             "avg_launch_us": round(fc1_s * 1e6, 2)}}
        model.set_precision(args.precision)

    # parity of what was just timed: attention-map L_inf vs the CPU oracle on the FIRST, MIDDLE and LAST tile of the bench
    # batch (a remap or batch-stride slip on later images must show), for the random-init weights of the timed run and for
    # the sharp (qkv x4) and peaked (attention max 0.8-0.9, what a trained checkpoint looks like: qkv x8 on ViT-S/16, the
    # geometry's calibrated gain otherwise - synth.stress_variant) weight sets pushed through the same engine at the same batch
    from oracle import vit_oracle as O  # checker
    pick = sorted({0, B // 2, B - 1})
    linf_by_set = {}
    for variant in ("init", "sharp", "peaked"):
        wset = synth.stress_variant(args.arch, p) if variant == "peaked" else variant
        sd = synth.synth_arch_state_dict(args.arch, p, seed=0, variant=wset)
        cfg = O.make_cfg(sd, p, H)
        ref = O.get_last_selfattention(sd, cfg, x[pick].cpu())
        if variant == "init":
            got = out["attn"][0][pick]
        else:
            model.load_state_dict(sd)
            got = model._run(x, flags=flags)["attn"][0][pick]
        linf_by_set[variant] = {"linf": float((got.cpu() - ref).abs().max()), "attn_max": round(float(ref.max()), 4),
                                "tiles_checked": pick, "weights": wset}
    linf = linf_by_set["init"]["linf"]
    if fast is not None:  # the peaked set's reference was computed in the loop above
        sdp = synth.synth_arch_state_dict(args.arch, p, seed=0, variant="peaked")
        refp = O.get_last_selfattention(sdp, O.make_cfg(sdp, p, H), x[pick].cpu())
        fast["attn_linf_peaked_weights"] = float((ob_pick - refp).abs().max())

    # HBM bytes per launch of the dominant class: measured offline with rocprofv3 --pmc (separate passes,
    # gfx950 FETCH_SIZE correction) and committed under profiles/; only valid for the default workload
    traffic = None
    if (args.arch, p, S, B) == ("vit_small", 16, 224, 64):
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[args.precision][dom]["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            traffic = None
    tiles = B * world * args.steps
    value = tiles / dt
    dom_avg_s = dom_ms / max(dom_n, 1) * 1e-3
    achieved = cf[dom] / dom_avg_s / 1e12 if dom_n else None
    peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_DENSE_TFLOPS
    # MFMA instructions issued per algorithmic product: split-bf16 evaluates hi*hi + hi*lo + lo*hi
    mfma_per_product = 3 if args.precision == "bf16x3" else 1
    line = {
        "metric": "OCM tiles/s (224x224, ViT-S/16 attention-map inference)" if (args.arch, p, S) == ("vit_small", 16, 224)
        else f"OCM tiles/s ({S}x{S}, {args.arch}/{p} attention-map inference)",
        "value": round(value, 1),
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "per_rank_ms": spread(rank_ms),
        "comm": comm,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"bf16": "bf16", "bf16x3": "bf16x3", "fp32": "f32"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"{args.arch} patch {p}, {B} tiles/GPU of {S}x{S} (RGB-replicated grayscale), "
                               f"get_last_selfattention -> (B,{H},{N},{N}) fp32 + CLS-row maps; random-init weights",
                   "tiles_per_gpu": B, "tokens": N, "parallelism": f"tile-shard x{world}" + (" + all-gather" if world > 1 else "")},
        "attn_linf_vs_cpu_oracle": linf,
        "attn_linf_by_weight_set": linf_by_set,
        "path_tflops": round(value * fmap / 1e12, 2),
        "path_frac_of_mfma_peak": round(value * fmap / 1e12 / (peak * world), 4),
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2) if achieved else None,
                     "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4) if achieved else None,
                     "traffic": traffic, "launches": int(dom_n), "avg_launch_us": round(dom_avg_s * 1e6, 2),
                     "flop_per_launch": cf[dom], "mfma_per_product": mfma_per_product,
                     "mfma_pipe_frac": round(achieved * mfma_per_product / peak, 4) if achieved else None},
        "kernel_breakdown": breakdown,
    }
    if fast is not None:
        line["fast_mode_bf16"] = fast
    if slab is not None:
        line["slab_sweep"] = slab
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline((D, L, H), p, S, B)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
