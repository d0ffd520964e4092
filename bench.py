#!/usr/bin/env python3
"""bench.py - denoising steps/sec of the PhotoVerse hot loop on MI355X.

One "step" = one iteration of the reference loop (/root/reference/models/infer.py:98-119): UNet(uncond) + UNet(cond)
at batch B, CFG combine, DPM-Solver++ update.  Workload (BASELINE.json configs[1]): SD-v1.5-shaped random-init UNet
with PhotoVerse cross-attention processors, B=16 per GPU, 64x64 latents (512x512), fp16 storage / fp32 accumulate,
synthetic conditioning.  N>1: one process per GPU, batch sharded (weak scaling), one all_gather of the final latents.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# analytic algorithmic FLOPs of one SD-v1.5 UNet forward per sample at 64x64 latents, P=1
UNET_TFLOP_PER_SAMPLE_64 = 0.8040
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
DOMINANT_KERNEL = "gemm_conv_kernel<5, 2, true, false, true>"   # as rocprofv3 prints it (tags in photoverse_amd/ops.py)


def cpu_baseline(seconds_budget=30.0):
    """The oracle (fp32 eager restatement of the reference path) timed on the host cores: B=1, one denoising step
    (2 UNet forwards + CFG + scheduler step).  Reported in bs=16-equivalent steps/s (measured B=1 rate / 16)."""
    import torch
    from oracle.infer_ref import denoise_ref
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    torch.manual_seed(0)
    t0 = time.time()
    unet = UNet2DConditionModelRef().eval()
    set_visual_cross_attention_adapter_ref(unet, (5,))
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(1, 4, 64, 64, generator=g)
    cond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    uncond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    build_s = time.time() - t0
    t0 = time.time()
    denoise_ref(unet, noise, cond, uncond, guidance_scale=7.5, timesteps=1)
    dt = time.time() - t0
    return {"value": (1.0 / dt) / 16.0, "unit": "denoising steps/s (bs=16 equivalent)", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"oracle fp32 eager, B=1 (1/16 of the batch), 1 step = 2 UNet fwd + CFG + DPM step, "
            f"{dt:.2f} s measured (+{build_s:.1f} s model build, untimed); value = (1/{dt:.2f})/16"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--ip-tokens", type=int, default=1, help="image tokens per sample (reference inference default: 1)")
    ap.add_argument("--guidance", type=float, default=7.5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batch-splits", type=int, default=1, help="independent sub-batches per forward (extra parallel graph branches)")
    ap.add_argument("--one-stream", action="store_true",
                    help="run the uncond / cond forwards back to back on one stream instead of as two parallel graph branches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ            # launched by torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from photoverse_amd.pipeline import DenoiseLoop, gather_latents, shard_batch
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter

    torch.manual_seed(0)                       # identical random-init weights on every rank
    unet = UNet2DConditionModel()
    set_visual_cross_attention_adapter(unet, (args.ip_tokens,))
    unet.to(dev)

    B, S, P, T = args.batch, args.latent, args.ip_tokens, max(args.steps, 1)
    loop = DenoiseLoop(unet, B, S, P, T, args.guidance, use_graph=not args.no_graph, two_streams=not args.one_stream, batch_splits=args.batch_splits)
    g = torch.Generator().manual_seed(1234)    # global batch drawn once on CPU (infer.py:52-59), sliced per rank
    GB = B * world
    noise = torch.randn(GB, 4, S, S, generator=g)
    text_c, text_u = torch.randn(GB, 77, 768, generator=g), torch.randn(GB, 77, 768, generator=g)
    ip_c, ip_u = torch.randn(GB, P, 768, generator=g), torch.randn(GB, P, 768, generator=g)
    sl = shard_batch(GB, rank, world)
    loop.set_conditioning((text_c[sl].to(dev), ip_c[sl].to(dev)), (text_u[sl].to(dev), ip_u[sl].to(dev)))
    loop.reset(noise[sl])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loop.step()
    torch.cuda.synchronize()
    loop.reset(noise[sl])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step()
    barrier()
    dt = time.perf_counter() - t0
    final = gather_latents(loop.latents, world, force=use_dist)      # the single collective of the path
    torch.cuda.synchronize()
    if use_dist:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    finite = bool(torch.isfinite(final).all().item())

    roofline = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel = the implicit-GEMM 3x3 conv instantiation (160-column tile, GroupNorm column statistics in the
        # epilogue: every 3x3 conv of the 64x64 / 32x32 levels); replay exactly its launches of one step and
        # time them with HIP events on the launch stream
        dom = DOMINANT_KERNEL
        subs = [e.rec.subset(lambda t: t[0] == dom) for e in loop.engines_u + loop.engines_c]
        nl = sum(len(s) for s in subs)
        flops = sum(t[1] for s in subs for t in s.tags)
        stream = torch.cuda.current_stream()
        for s in subs:
            s.run()
        torch.cuda.synchronize()
        reps = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            for s in subs:
                s.run()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ach = flops / (ms * 1e-3) / 1e12
        # HBM traffic of this kernel from the rocprofv3 PMC passes of the same command (tools/profile_bench.py; counters are
        # collected in their own runs, so the number is read from the committed summary, not measured in this process)
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_d_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            traffic, traffic_src = pmc.get("dominant_hbm_bytes_per_launch"), "profiles/r01_d_pmc_traffic.json: " + pmc.get("note", "")
        except (OSError, ValueError):
            pass
        algo_bytes = sum(t[2] for s in subs for t in s.tags) / nl
        roofline = {"bound": "mfma", "kernel": DOMINANT_KERNEL, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "measured": "HIP events around a back-to-back replay of this kernel's launches of one step on one stream "
                                "(in the timed loop the uncond/cond forwards are two overlapping graph branches, so rocprof "
                                "per-dispatch durations of the default run include co-scheduling; `--one-stream` is the matching run)",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": algo_bytes, "launches_per_step": nl,
                    "avg_launch_us": round(ms * 1e3 / nl, 2), "flops_per_launch_avg": flops / nl,
                    "share_of_step_flops": round(flops / (2 * B * UNET_TFLOP_PER_SAMPLE_64 * (S / 64) ** 2 * 1e12), 3),
                    "real_data_ceiling": {"value": 1700.0, "unit": "TFLOP/s", "frac": round(ach / 1700.0, 4),
                                          "source": "profiles/r01_mfma_power_probe.txt: this kernel's MFMA + ds_read mix on random fp16 "
                                                    "operands, no global traffic, holds 1.70 GHz (2.40 GHz / 2.45 PFLOP/s only with zeros)"}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (contract); N > 1 runs stay short
        cpu = cpu_baseline()

    if rank == 0:
        value = world * args.steps / dt
        step_tflop = 2 * B * UNET_TFLOP_PER_SAMPLE_64 * (S / 64) ** 2 if S == 64 else None
        out = {
            "metric": "denoising steps/sec at 512x512 bs=16, 50-step loop (2 UNet fwd + CFG + DPM-Solver++ step per step)",
            "value": round(value, 3), "unit": "denoising steps/s (bs=16 per GPU, summed over GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random-init SD-v1.5-shaped weights, N(0,1) latents / text / image-token embeddings)",
            "config": {"workload": "configs[1]: SD-v1.5 UNet + PhotoVerse processors, 50-step loop, bs=16/GPU, 512x512 (64x64 latents), fp16",
                       "per_gpu_batch": B, "global_batch": GB, "latent": S, "ip_tokens": P, "guidance_scale": args.guidance,
                       "parallelism": f"dp{world} (batch-sharded, 1 all_gather)", "hip_graph": not args.no_graph, "graph_branches": 1 if args.one_stream else 2 * args.batch_splits,
                       "launches_per_step": loop.launches_per_step},
            "finite": finite,
            "step_mfma_frac": (round(step_tflop / (dt / args.steps) / 1e0 / MFMA_PEAK_TFLOPS, 4) if step_tflop else None),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()               # rank 0's roofline replay is over: every rank leaves the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
