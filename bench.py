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
#: committed PMC summary `roofline.traffic` is read from (tools/profile_bench.py writes it together with the git blob hashes of the
#: kernel sources it was measured on; bench.py reports `traffic_stale` when those differ from the sources in the tree)
PMC_TRAFFIC_FILE = "profiles/r06_pmc_traffic.json"
KERNEL_SOURCES = ("photoverse_amd/csrc/pv_gemm.hip", "photoverse_amd/csrc/pv_convbig.hip", "photoverse_amd/csrc/pv_attn.hip")


def git_blob_sha1(path):
    """`git hash-object` of a file without git (the GPU box has no .git): sha1(b"blob <len>\0" + content)."""
    import hashlib
    with open(path, "rb") as fh:
        data = fh.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def cpu_baseline(seconds_budget=40.0):
    """The oracle (fp32 eager restatement of the reference path) timed on the host cores: B=1, one denoising step
    (2 UNet forwards + CFG + scheduler step).  One untimed warm-up step, then 2 timed steps and a third while the budget lasts;
    the MEDIAN is reported in bs=16-equivalent steps/s (measured B=1 rate / 16)."""
    import statistics
    import torch
    from oracle.infer_ref import denoise_ref
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    torch.manual_seed(0)
    t_begin = time.time()
    unet = UNet2DConditionModelRef().eval()
    set_visual_cross_attention_adapter_ref(unet, (5,))
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(1, 4, 64, 64, generator=g)
    cond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    uncond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    build_s = time.time() - t_begin
    t0 = time.time()
    denoise_ref(unet, noise, cond, uncond, guidance_scale=7.5, timesteps=1)          # warm-up (allocator, thread pool)
    warm_s = time.time() - t0
    times = []
    while len(times) < 3 and (len(times) < 2 or (time.time() - t_begin) + warm_s < seconds_budget):      # never fewer than two timed samples
        t0 = time.time()
        denoise_ref(unet, noise, cond, uncond, guidance_scale=7.5, timesteps=1)
        times.append(time.time() - t0)
    dt = statistics.median(times)
    return {"value": (1.0 / dt) / 16.0, "unit": "denoising steps/s (bs=16 equivalent)", "cores": torch.get_num_threads(),
            "kind": "port", "samples": len(times),
            "sample": f"oracle fp32 eager, B=1 (1/16 of the batch), 1 step = 2 UNet fwd + CFG + DPM step; 1 warm-up step "
                      f"({warm_s:.2f} s) + {len(times)} timed: {', '.join('%.2f' % t for t in times)} s, median {dt:.2f} s "
                      f"(+{build_s:.1f} s model build, untimed); value = (1/{dt:.2f})/16"}


def build_random_unet(ip_tokens, dev):
    """SD-v1.5-shaped UNet + PhotoVerse processors with seeded random weights, drawn ON THE DEVICE: torch's default CPU initialisers take
    ~30 s for the 878 M parameters (the driver's clock runs around the whole script); U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like
    nn.Linear / nn.Conv2d defaults, norm scales 1 / biases 0, identical on every rank (same seed)."""
    import torch
    import torch.nn as nn
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    saved = [(cls, cls.reset_parameters) for cls in (nn.Linear, nn.Conv2d, nn.GroupNorm, nn.LayerNorm)]
    for cls, _ in saved:
        cls.reset_parameters = lambda self: None
    try:
        unet = UNet2DConditionModel()
        set_visual_cross_attention_adapter(unet, (ip_tokens,))
    finally:
        for cls, fn in saved:
            cls.reset_parameters = fn
    gen = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for name, p_ in unet.named_parameters():
            q = torch.empty(p_.shape, dtype=p_.dtype, device=dev)
            if p_.ndim >= 2:
                bound = 1.0 / (p_[0].numel() ** 0.5)
                q.uniform_(-bound, bound, generator=gen)
            elif name.endswith("weight"):
                q.fill_(1.0)
            else:
                q.zero_()
            p_.data = q
    unet.to(dev)          # packs the fp16 operands of the engines
    return unet


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n, argv, dry=False):
    """`python bench.py --gpus N` without an external launcher: the parent starts N child processes of this script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (one rank per GPU over RCCL), relays rank 0's JSON line and exits with
    the worst child code.  The parent never initialises the GPU and never exec()s (a process that has touched the GPU must
    not be replaced on this pool); `torch.cuda.device_count()` does not initialise it."""
    import subprocess
    if not dry:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < n:
            print(f"bench.py: --gpus {n} requested but only {ndev} HIP device(s) are visible; nothing was run", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: required for RCCL across processes on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return max(abs(c) for _, c in bad) or 1
    return 0


def train_step_leg(unet, B, S, dev, reps=3, with_face=True):
    """Time one full training iteration at configs[3]'s shape: bs=16, 64x64 latents, 5 image tokens, LoRA (r=8, dropout 0.1) on attn2.to_q/k/v,
    SD-v1.5-sized UNet, 12-layer CLIP text encoder, both adapters; forward + backward + clip + AdamW."""
    import time
    import torch
    from photoverse_amd.adapters import PhotoVerseAdapter
    from photoverse_amd.clip import CLIPTextModel
    from photoverse_amd.lora import LoraConfig, inject_adapter_in_model
    from photoverse_amd.optim import AdamW
    from photoverse_amd.train import TrainStep
    torch.manual_seed(7)
    inject_adapter_in_model(LoraConfig(r=8, lora_alpha=1, lora_dropout=0.1), unet)      # train.py:258-275 defaults
    unet.to(dev)
    for m in unet.modules():
        if hasattr(m, "lora_B"):
            m.lora_B["default"].weight.data.normal_(0, 0.02)
    text_encoder = CLIPTextModel().to(dev)
    text_adapter = PhotoVerseAdapter(1024, 768, 5).to(dev)
    image_adapter = PhotoVerseAdapter(1024, 768, 5).to(dev)
    t0 = time.perf_counter()
    ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=S, w=S, n_tokens=5, grad_scale=4096.0, fusion_seed=1)
    build_s = time.perf_counter() - t0
    groups = ts.trainable_parameters()
    opt = AdamW([p for g_ in groups.values() for p in g_], lr=1e-5, weight_decay=1e-2)
    g = torch.Generator().manual_seed(99)
    inputs = dict(noisy_latents=torch.randn(B, 4, S, S, generator=g).to(dev), noise=torch.randn(B, 4, S, S, generator=g).to(dev),
                  timesteps=torch.randint(0, 1000, (B,), generator=g), text_input_ids=torch.randint(0, 49000, (B, 77), generator=g).to(dev),
                  placeholder_idx=torch.full((B, 1), 5).to(dev),
                  image_embeddings=[torch.randn(B, 257, 1024, generator=g).half().to(dev) for _ in range(5)])

    def one():
        out = ts.step(**inputs)
        opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=ts.grad_scale)
        return out
    out = one()                                              # warm-up (first-launch attribute setup, allocator)
    torch.cuda.synchronize()
    l0 = float(out["loss"])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ts.tape.load_weights()                                   # the two plans eagerly, timed separately
    ev[0].record()
    ts.tape.rf.run()
    ev[1].record()
    ts.tape.rb.run()
    ev[2].record()
    torch.cuda.synchronize()
    fwd_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    out = one()                                              # captures the HIP graph
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        out = one()
    torch.cuda.synchronize()
    iter_ms = (time.perf_counter() - t1) * 1e3 / reps
    hip_graph = ts.graph is not None
    launches = (len(ts.tape.rf), len(ts.tape.rb))
    # roofline of the iteration: algorithmic flops of every tagged launch of the two plans (GEMM / conv 2 M N K - the data gradients are the
    # same contractions on transposed weights -, attention 4 N Nk d per head forward and 10 N Nk d backward (five products), weight
    # gradients 2 m n k) over the graph-replayed iteration time; and the backward kernel with the largest share of them, timed alone
    fl_f, fl_b = sum(t[1] for t in ts.tape.rf.tags), sum(t[1] for t in ts.tape.rb.tags)
    by_name = {}
    for t in ts.tape.rb.tags:
        by_name[t[0]] = by_name.get(t[0], 0.0) + t[1]
    dom = max(by_name, key=by_name.get)
    sub = ts.tape.rb.subset(lambda t: t[0] == dom)
    sub.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for _ in range(3):
        sub.run()
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    dom_ms = e0.elapsed_time(e1) / 3
    train_roofline = {"bound": "mfma", "algorithmic_tflop": {"forward_plan": round(fl_f / 1e12, 3), "backward_plan": round(fl_b / 1e12, 3)},
                      "achieved": round((fl_f + fl_b) / (iter_ms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round((fl_f + fl_b) / (iter_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                      "forward_plan_frac": round(fl_f / (fwd_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                      "backward_plan_frac": round(fl_b / (bwd_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                      "dominant_backward_kernel": {"kernel": dom, "launches": len(sub), "share_of_backward_flops": round(by_name[dom] / fl_b, 3),
                                                   "avg_launch_us": round(dom_ms * 1e3 / len(sub), 2),
                                                   "achieved": round(by_name[dom] / (dom_ms * 1e-3) / 1e12, 1),
                                                   "frac": round(by_name[dom] / (dom_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                                   "measured": "HIP events around a back-to-back replay of this kernel's launches of one backward plan"},
                      "flops_counted": "2 M N K per GEMM / conv launch (forward, data gradient, weight gradient), 4 B H N Nk d per attention forward, "
                                       "10 B H N Nk d per attention backward; norms / activations / optimizer excluded"}
    act_bytes = ts.tape.rf.bytes_allocated + ts.tape.rb.bytes_allocated
    loss_last = float(out["loss"])
    finite = bool(torch.isfinite(out["loss"]).all().item())
    # the same iteration WITH the identity-loss branch (configs[3] as written: "... with ArcFace identity loss"): 4 of the 16 samples
    # (face_loss_sample_ratio 0.25) go through run_inference(timesteps=10, guidance 2, training_mode=True) - 9 denoising steps without
    # gradient, the last one + the VAE decode (512x512) + the IR-ResNet18 inside the differentiated plan
    face = None
    if with_face:
        from photoverse_amd.loss import FaceLoss
        from photoverse_amd.vae import AutoencoderKL
        from photoverse_amd.modeling_utils import load_models  # noqa: F401
        from types import SimpleNamespace
        from photoverse_amd.scheduler import DPMSolverMultistepScheduler
        del ts, opt
        torch.cuda.empty_cache()
        vae = AutoencoderKL().to(dev)
        fl = FaceLoss(dev, "arcface")
        ns = 4
        ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=S, w=S, n_tokens=5, grad_scale=4096.0, fusion_seed=1,
                       face_loss=fl, vae=vae, noise_scheduler=SimpleNamespace(config=DPMSolverMultistepScheduler().config), face_samples=ns,
                       guidance_scale=2.0, infer_steps=10)
        groups = ts.trainable_parameters()
        opt = AdamW([p for g_ in groups.values() for p in g_], lr=1e-5, weight_decay=1e-2)
        fi = dict(pixel_values=(torch.rand(ns, 3, 8 * S, 8 * S, generator=g) * 2 - 1).to(dev), start_latents=torch.randn(ns, 4, S, S, generator=g).to(dev),
                  image_embeddings=torch.randn(ns, 257, 1024, generator=g).half().to(dev),
                  uncond_image_embeddings=torch.randn(ns, 257, 1024, generator=g).half().to(dev),
                  text_input_ids=torch.randint(0, 49000, (ns, 77), generator=g).to(dev), placeholder_idx=torch.full((ns, 1), 4).to(dev),
                  uncond_input_ids=torch.randint(0, 49000, (ns, 77), generator=g).to(dev))

        def one_face():
            o = ts.step(**inputs, face_inputs=fi)
            opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=ts.grad_scale)
            return o
        o = one_face()
        o = one_face()                                       # second iteration captures the HIP graphs
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            o = one_face()
        torch.cuda.synchronize()
        fms = (time.perf_counter() - t1) * 1e3 / 2
        # where the branch's time goes: its four launch plans replayed alone between events (eager launches on the current stream)
        fb = ts.face

        def plan_ms(rec, reps, state0):
            fb.state.copy_(torch.tensor([state0, fb.T, 0, 0], dtype=torch.int32))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rec.run()
            fb.state.copy_(torch.tensor([state0, fb.T, 0, 0], dtype=torch.int32))
            e0.record()
            for _ in range(reps):
                rec.run()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / reps, 2)
        pieces = {"conditioning": plan_ms(fb.rec_cond, 2, 0), "no_grad_step": plan_ms(fb.loop_tape.rf, 4, 0),
                  "last_step_decode_loss": plan_ms(fb.rec_last, 2, fb.T - 1), "backward": plan_ms(fb.tape.rb, 2, fb.T - 1)}
        face = {"ms_per_iteration": round(fms, 2), "plans_ms": pieces, "face_samples": ns, "infer_steps": 10, "guidance_scale": 2.0,
                "face_loss": round(float(o["face_loss"]), 5), "loss": round(float(o["loss"]), 5), "finite": bool(torch.isfinite(o["loss"]).all().item()),
                "launches": {"conditioning": len(ts.face.rec_cond), "no_grad_step": len(ts.face.loop_tape.rf), "last_step_decode_loss": len(ts.face.rec_last),
                             "backward": len(ts.face.tape.rb)}}
    n_train = sum(p.numel() for g_ in groups.values() for p in g_)
    return {"workload": "configs[3]: bs=16, 64x64 latents, 5 image tokens, LoRA r=8 / alpha=1 / dropout=0.1 (the reference defaults) on attn2.to_q/k/v; "
                        "adapters + 12-layer CLIP text encoder + SD-v1.5 UNet forward, backward through all of them, per-module clip_grad_norm_, AdamW; "
                        "ms_per_iteration etc. are WITHOUT the ArcFace term, with_face_loss is the same iteration with it",
            "ms_per_iteration": round(iter_ms, 2), "forward_plan_ms": round(fwd_ms, 2), "backward_plan_ms": round(bwd_ms, 2), "roofline": train_roofline,
            "launches_forward": launches[0], "launches_backward": launches[1], "trainable_parameters": n_train,
            "activation_bytes": act_bytes, "plan_build_s": round(build_s, 2),
            "loss_first": round(l0, 5), "loss_last": round(loss_last, 5), "finite": finite,
            "hip_graph": hip_graph, "samples_per_s": round(B / (iter_ms * 1e-3), 2), "with_face_loss": face}


def workload_label(B, S, P, guidance, steps, world):
    """Which BASELINE.json config the arguments describe (B = per-GPU batch).  configs[1]: bs=16, 64x64 latents, P=1, 50 steps on one GPU;
    configs[2]: the same per GPU on 8 GPUs (bs=128, batch-sharded); configs[4]: bs=32 over 8 GPUs = 4 per GPU, 96x96 latents (768x768),
    P = 6 image tokens (extra_num_tokens=16 widens the adapter, not the token count of a layer: SURVEY 8d).  Anything else says so."""
    shape = f"SD-v1.5 UNet + PhotoVerse processors, {steps}-step loop, bs={B}/GPU, {8 * S}x{8 * S} ({S}x{S} latents), P={P}, guidance {guidance:g}, fp16"
    if (B, S, P, guidance) == (16, 64, 1, 7.5):
        if steps != 50:
            head = f"configs[1] shape timed over {steps} steps instead of 50" + ("" if world == 1 else f", per GPU x {world} GPUs (configs[2]-shaped, batch-sharded)")
        elif world == 1:
            head = "configs[1]"
        elif world == 8:
            head = "configs[2]: bs=128 batch-sharded over 8 GPUs (configs[1] per GPU), one all-gather of the final latents"
        else:
            head = f"configs[2]-shaped (configs[1] per GPU x {world} GPUs, batch-sharded)"
    elif (B, S, P, guidance) == (4, 96, 6, 7.5):
        head = ("configs[4] per-rank shape" if world == 1 else
                "configs[4]: bs=32 batch-sharded over 8 GPUs" if world == 8 else f"configs[4]-shaped (its per-rank shape x {world} GPUs)")
    else:
        head = "custom shape (NOT a BASELINE config)"
    return head + ": " + shape


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
    ap.add_argument("--no-train-forward", action="store_true", help="skip the separately labelled training-shaped forward measurement")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher self-test (no GPU): every rank reports its RANK / WORLD_SIZE and exits")
    args = ap.parse_args()

    # ---- launch: N ranks, one per GPU --------------------------------------------------------------------------
    # Under an external launcher (torch.distributed.run sets RANK / WORLD_SIZE) this process IS one rank; otherwise the
    # parent spawns the N ranks itself, BEFORE anything touches the GPU.
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], dry=args.dry_launch))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_launch:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            seen = [None] * world
            dist.all_gather_object(seen, (rank, local_rank, os.getpid()))
            # the data path's one collective, on CPU tensors: every rank contributes its shard of a global batch drawn once and sliced
            # (pipeline.shard_batch), one all_gather_into_tensor puts the shards back in rank order
            import torch
            from photoverse_amd.pipeline import shard_batch
            gb = args.batch * world
            glob = torch.arange(gb * 4, dtype=torch.float32).view(gb, 4)
            got = torch.empty_like(glob)
            dist.all_gather_into_tensor(got, glob[shard_batch(gb, rank, world)].contiguous())
            gather_ok = bool(torch.equal(got, glob))
            dist.destroy_process_group()
        else:
            seen, gather_ok = [(rank, local_rank, os.getpid())], True
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": [s_[0] for s_ in seen],
                              "local_ranks": [s_[1] for s_ in seen], "distinct_pids": len({s_[2] for s_ in seen}), "gather_in_rank_order": gather_ok,
                              "global_batch": args.batch * world,
                              "workload": workload_label(args.batch, args.latent, args.ip_tokens, args.guidance, args.steps, world)}))
        return

    import torch
    import torch.distributed as dist

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
    from photoverse_amd.unet import set_visual_cross_attention_adapter

    torch.manual_seed(0)
    unet = build_random_unet(args.ip_tokens, dev)          # identical random-init weights on every rank

    B, S, P, T = args.batch, args.latent, args.ip_tokens, max(args.steps, args.warmup, 1)   # the schedule covers the warm-up too
    loop = DenoiseLoop(unet, B, S, P, T, args.guidance, use_graph=not args.no_graph, two_streams=not args.one_stream, batch_splits=args.batch_splits,
                       share_prefix=False)       # the headline: two FULL forwards per step
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
    rank_ms = [dt / args.steps * 1e3]
    if use_dist:
        # every rank contributes its own clock: the line reports min / max over ranks, `value` uses the MAX (contract)
        mine = torch.tensor([dt], device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        rank_ms = [t.item() / args.steps * 1e3 for t in allt]
        dt = max(t.item() for t in allt)
    rccl_world = dist.get_world_size() if use_dist else 1
    finite = bool(torch.isfinite(final).all().item())

    def replay(groups, reps=5):
        """``groups``: one after the other; a group is a list of LANES, a lane a list of recorders run in order on one stream: lane 0 on the current
        stream, the others on a side stream, concurrently - a step's schedule (the uncond and cond plans side by side, the merged low-resolution plan alone)
        restricted to the launches in the recorders.  Returns ms per replay (HIP events on the launch stream)."""
        stream = torch.cuda.current_stream()
        sides = [torch.cuda.Stream(device=dev) for _ in range(max(len(g_) for g_ in groups) - 1)]

        def once():
            for grp in groups:
                for lane, sd in zip(grp[1:], sides):
                    sd.wait_stream(stream)
                    with torch.cuda.stream(sd):
                        for r_ in lane:
                            r_.run()
                for r_ in grp[0]:
                    r_.run()
                for _, sd in zip(grp[1:], sides):
                    stream.wait_stream(sd)
        once()
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            once()
        b_.record(stream)
        torch.cuda.synchronize()
        return a.elapsed_time(b_) / reps

    def step_schedule(subs_of):
        """The replay groups of ``subs_of`` (engine -> recorder subset) in the step's own schedule."""
        lane_u = [subs_of[e] for e in loop.engines_u if len(subs_of[e])]
        lane_c = [subs_of[e] for e in loop.engines_c if len(subs_of[e])]
        alone = [[[subs_of[e]]] for e in loop.engines_m if len(subs_of[e])]
        if args.one_stream or not lane_u or not lane_c:
            return [[[r_]] for r_ in lane_u + lane_c] + alone
        return [[lane_u, lane_c]] + alone

    roofline = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel = the launch SYMBOL with the largest share of one step's algorithmic flops.  Rounds 1-4 that was the 3x3 conv on the big tile
        # (25 % of a step's flops under one symbol); since round 5 the 64 x 64 convs run on the LDS-resident-patch instantiation (MODE 3, 13.2 %) and the
        # 32 x 32 convs stay on the gathered one (MODE 0, 12.2 %), so the largest single symbol is the d = 40 self-attention kernel attn8_kernel<497>
        # (13.4 % of the flops, ~15 % of the time) and the patch conv is the runner-up.  Both get the same measurement: exactly the symbol's launches
        # of one step, replayed and timed with HIP events on the launch stream.
        engines = loop.all_engines
        by_kernel = {}
        for e in engines:
            for t in e.rec.tags:
                by_kernel[t[0]] = by_kernel.get(t[0], 0.0) + t[1]
        ranked = sorted(by_kernel, key=by_kernel.get, reverse=True)

        def kernel_roofline(sym):
            subs = [e.rec.subset(lambda t: t[0] == sym) for e in engines]
            nl = sum(len(s_) for s_ in subs)
            flops = sum(t[1] for s_ in subs for t in s_.tags)
            # as the step schedules it: the two per-branch plans side by side (their big-tile launches may have 128 one-per-CU workgroups each: half the
            # chip, by design - pv_gemm_params.big_tile_min), the merged low-resolution plan alone
            ms = replay(step_schedule(dict(zip(engines, subs))))
            ach = flops / (ms * 1e-3) / 1e12
            # the same kernel one launch at a time (what a rocprofv3 per-dispatch duration of an un-overlapped launch shows), over its chip-filling launches
            # only (>= 256 workgroups)
            single = None
            big_subs = [e.rec.subset(lambda t: t[0] == sym and len(t) > 3 and t[3] >= 256) for e in engines]
            nb = sum(len(s_) for s_ in big_subs)
            if nb:
                fb = sum(t[1] for s_ in big_subs for t in s_.tags)
                msb = replay([[[r_]] for r_ in big_subs if len(r_)])
                single = {"what": "launches of this kernel with >= 256 workgroups (they fill the chip alone), replayed one at a time on one stream", "launches_per_step": nb,
                          "avg_launch_us": round(msb * 1e3 / nb, 2), "achieved": round(fb / (msb * 1e-3) / 1e12, 1), "frac": round(fb / (msb * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
            return {"kernel": sym, "achieved": round(ach, 1), "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "single_launch": single, "launches_per_step": nl,
                    "avg_launch_us": round(ms * 1e3 / nl, 2), "flops_per_launch_avg": flops / nl,
                    "algorithmic_bytes_per_launch": sum(t[2] for s_ in subs for t in s_.tags) / nl,
                    "share_of_step_flops": round(flops / (2 * B * UNET_TFLOP_PER_SAMPLE_64 * (S / 64) ** 2 * 1e12), 3)}

        dom = ranked[0]
        first = kernel_roofline(dom)
        second = kernel_roofline(ranked[1]) if len(ranked) > 1 else None
        # HBM traffic of the dominant kernel from the rocprofv3 PMC passes of the same command (tools/profile_bench.py; counters are
        # collected in their own runs, so the number is read from the committed summary, not measured in this process)
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as fh:
                pmc = json.load(fh)
            if pmc.get("dominant_kernel") == dom:
                traffic, traffic_src = pmc.get("dominant_hbm_bytes_per_launch"), PMC_TRAFFIC_FILE + ": " + pmc.get("note", "")
                now = {k: git_blob_sha1(os.path.join(ROOT, k)) for k in KERNEL_SOURCES}
                traffic_stale = pmc.get("source_blobs") != now       # the kernel sources changed since the counters were collected
            else:
                traffic_src = f"{PMC_TRAFFIC_FILE} holds {pmc.get('dominant_kernel')!r}, not this kernel: no traffic figure"
        except (OSError, ValueError):
            pass
        roofline = dict(first)
        roofline.update({"bound": "mfma", "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "measured": "HIP events around a replay of this kernel's launches of one step AS THE STEP SCHEDULES THEM: the launches of the uncond and cond "
                                "plans on two streams side by side, the merged low-resolution plan's alone.  achieved = algorithmic flops / elapsed; avg_launch_us = elapsed / "
                                "launches (rocprofv3 per-dispatch durations of overlapping dispatches are wall durations: their sum exceeds the elapsed time by the "
                                "overlap).  `single_launch` is the same kernel one launch at a time; `runner_up` the symbol with the next-largest share of the flops, same measurement",
                    "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                    "runner_up": second,
                    "real_data_ceiling": {"value": 1700.0, "unit": "TFLOP/s", "frac": round(first["achieved"] / 1700.0, 4),
                                          "source": "profiles/r01_mfma_power_probe.txt: an MFMA + ds_read mix on random fp16 "
                                                    "operands, no global traffic, holds 1.70 GHz (2.40 GHz / 2.45 PFLOP/s only with zeros)"}})

    # the north_star's named target: MFMA utilisation of the adapter cross-attention kernel.  SURVEY 0.1 #8 defines it as the fused
    # to_q + dual-branch SDPA + to_out kernel.  It exists for the C = 320 layers (pv_xfused.hip); the 640 / 1280-channel layers run four
    # launches (LayerNorm, to_q GEMM, dual-branch attention, to_out GEMM + residual).  Reported per level AND over all 16 attn2 layers,
    # time-weighted, against the 0.40 target.  Same measurement as the roofline object: the branch's launches of one step replayed
    # in the step's schedule (the uncond and cond plans side by side on two streams, the merged low-resolution plan alone) between HIP events.  (The text / image-token K, V projections depend on the conditioning only and run
    # once per generation, outside the step.)
    xfused = None
    if rank == 0 and not args.no_roofline:
        levels, tot_ms, tot_fl, tot_n = {}, 0.0, 0.0, 0
        for C_ in (320, 640, 1280):
            subs = [e.rec.subset_role(f"attn2:{C_}") for e in loop.all_engines]
            nl = sum(len(s_) for s_ in subs)
            if not nl:
                continue
            flops = sum(t[1] for s_ in subs for t in s_.tags)
            kinds = sorted({t[0] for s_ in subs for t in s_.tags})
            ms = replay(step_schedule(dict(zip(loop.all_engines, subs))))
            # layer instances: one SDPA launch per layer and forward; the merged low-resolution plan runs BOTH CFG forwards' layers in one launch
            layers = sum((e.B // B) for e, s_ in zip(loop.all_engines, subs) for t in s_.tags
                         if t[0].startswith("xattn_fused_kernel") or t[0] in ("pv_cross_attention", "xattn_lnq_kernel"))
            levels[str(C_)] = {"layers_per_step": layers, "launches_per_step": nl, "launches_per_layer": sum((e.B // B) * len(s_) for e, s_ in zip(loop.all_engines, subs)) // max(layers, 1), "ms_per_step": round(ms, 4),
                               "us_per_layer": round(ms * 1e3 / max(layers, 1), 2), "achieved": round(flops / (ms * 1e-3) / 1e12, 1),
                               "frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4), "kernels": kinds,
                               "fused": any(k.startswith("xattn_fused_kernel") for k in kinds), "head_parallel": "xattn_lnq_kernel" in kinds}
            tot_ms, tot_fl, tot_n = tot_ms + ms, tot_fl + flops, tot_n + layers
        if levels:
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            xfused = {"what": "attn2 branch of every transformer block: norm2 -> to_q -> text + image-token SDPA (two softmaxes) -> to_out + bias + "
                              "residual; ONE launch (pv_cross_attention_fused) at C = 320 / d = 40 and C = 640 / d = 80; TWO at C = 1280 / d = 160: "
                              "norm2 + to_q + both SDPAs head-parallel (pv_cross_attention_lnq: 128 rows x one head per workgroup, 256 workgroups "
                              "on the 16 x 16 level) and to_out + bias + residual as a GEMM (round 3: four launches)",
                      "levels": levels, "all_layers": {"layers_per_step": tot_n, "ms_per_step": round(tot_ms, 4), "achieved": round(ach, 1),
                                                       "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "weighting": "sum of algorithmic flops / sum of time"},
                      "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "north_star_target_frac": 0.40,
                      "flops_counted": "algorithmic: 4 M C^2 (to_q + to_out) + 4 M (77 + P) C (both SDPA products), M = B * H * W of the level",
                      "phase_roofline": "profiles/r04_xfused_phase_roofline.md", "pmc": "profiles/r03_pmc_xfused.txt (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_BUSY_CYCLES of the fused kernel)"}

    # OPTIONAL, separately labelled (round-3 verdict item 9): the two CFG forwards of a step share the part that never sees the conditioning (conv_in ->
    # first ResnetBlock -> first transformer block up to attn1): exact (bit-identical latents), 2.5 % fewer flops.  `value` above stays two FULL forwards.
    shared = None
    if rank == 0 and world == 1 and not args.no_roofline and not args.one_stream and args.batch_splits == 1:
        k2 = min(args.steps, 20)
        loop2 = DenoiseLoop(unet, B, S, P, max(k2, 3), args.guidance, use_graph=not args.no_graph, share_prefix=True)
        if loop2.share_prefix:
            loop2.set_conditioning((text_c[sl].to(dev), ip_c[sl].to(dev)), (text_u[sl].to(dev), ip_u[sl].to(dev)))
            loop2.reset(noise[sl])
            for _ in range(3):
                loop2.step()
            torch.cuda.synchronize()
            loop2.reset(noise[sl])
            t2 = time.perf_counter()
            for _ in range(k2):
                loop2.step()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t2) / k2
            fl_full = sum(t[1] for e in loop.all_engines for t in e.rec.tags)
            fl2 = sum(t[1] for e in loop2.all_engines for t in e.rec.tags)
            shared = {"what": "NOT the headline: the same loop with the conditioning-independent prefix of the two CFG forwards (conv_in, first ResnetBlock, first transformer "
                              "block up to attn1) computed once per step instead of twice (DenoiseLoop(share_prefix=True)); latents bit-identical "
                              "(tests/test_unet_gpu.py::test_shared_prefix_of_the_two_cfg_forwards_is_exact)",
                      "value": round(1.0 / dt2, 3), "unit": "denoising steps/s (bs=%d)" % B, "ms_per_step": round(dt2 * 1e3, 3), "steps": k2,
                      "flops_vs_two_full_forwards": round(fl2 / fl_full, 4), "launches_per_step": loop2.launches_per_step,
                      "step_mfma_frac_of_reduced_flops": round(fl2 / 1e12 / dt2 / MFMA_PEAK_TFLOPS, 4)}
        loop2 = None
        torch.cuda.empty_cache()

    # second, separately labelled config (BASELINE configs[3], forward half only): the UNet forward a TRAINING step runs
    # (train.py:495-506) - P = 5 image tokens, per-sample timesteps, grad-mode branch fusion drawn on the device per layer.
    # Inference-style engine (fused GEGLU / fused attn2, no activations kept); the whole iteration is `train_step` below.  NOT part of `value`.
    train_fwd = ip5 = cfg4 = None
    launches_per_step = loop.launches_per_step
    if rank == 0 and world == 1 and not args.no_train_forward and S == 64:
        loop = subs = None          # release the loop's ~8 GB of static buffers before building the training-shaped engine
        torch.cuda.empty_cache()
        set_visual_cross_attention_adapter(unet, (5,))
        unet.to(dev)
        eng = unet.engine(B, S, S, 5, B, device_fusion="always", fusion_seed=1)
        gt = torch.Generator().manual_seed(99)
        eng.x_in.copy_(torch.randn(B, 4, S, S, generator=gt))
        eng.text.copy_(torch.randn(B * 77, 768, generator=gt))
        eng.ip.copy_(torch.randn(B * 5, 768, generator=gt))
        eng.timesteps.copy_(torch.randint(0, 1000, (B,), generator=gt).float())
        eng.run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eng.rec_cond.run()          # a training step re-projects the conditioning every iteration (adapters are being trained)
            eng.rec.run()
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        tfl = B * 0.8043
        train_fwd = {"workload": "configs[3], forward only: SD-v1.5 UNet + PhotoVerse processors, bs=16, 64x64 latents, P=5 image tokens, "
                                 "per-sample timesteps, grad-mode branch fusion drawn on the device (no host sync), K/V projections included",
                     "ms_per_forward": round(ms, 3), "tflop_per_forward": round(tfl, 3), "achieved_tflops": round(tfl / (ms * 1e-3), 1),
                     "mfma_frac": round(tfl / (ms * 1e-3) / MFMA_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(eng.out).all().item()),
                     "launches": len(eng.rec) + len(eng.rec_cond), "backward": "see train_step"}

        # configs[1] with ALL FIVE image tokens (token_index = 'full': SURVEY 8d asks for P = 1 and P = 5 both): the same loop, separately labelled
        eng = gr = None
        torch.cuda.empty_cache()
        k5 = min(args.steps, 15)
        loop5 = DenoiseLoop(unet, B, S, 5, max(k5, 3), args.guidance, use_graph=not args.no_graph, share_prefix=False)
        g5 = torch.Generator().manual_seed(4321)
        mk = lambda *sh: torch.randn(*sh, generator=g5).to(dev)
        loop5.set_conditioning((mk(B, 77, 768), mk(B, 5, 768)), (mk(B, 77, 768), mk(B, 5, 768)))
        n5 = torch.randn(B, 4, S, S, generator=g5)
        loop5.reset(n5)
        for _ in range(3):
            loop5.step()
        torch.cuda.synchronize()
        loop5.reset(n5)
        t5 = time.perf_counter()
        for _ in range(k5):
            loop5.step()
        torch.cuda.synchronize()
        dt5 = (time.perf_counter() - t5) / k5
        ip5 = {"workload": "configs[1] with P = 5 image tokens (token_index = 'full'): same loop, bs=%d, 64x64 latents, guidance %.1f" % (B, args.guidance),
               "value": round(1.0 / dt5, 3), "unit": "denoising steps/s (bs=%d)" % B, "ms_per_step": round(dt5 * 1e3, 3), "steps": k5,
               "finite": bool(torch.isfinite(loop5.latents).all().item())}
        loop5 = None
        torch.cuda.empty_cache()
        # BASELINE configs[4], ONE rank's share (total batch 32 over 8 ranks: 4 samples of 96 x 96 latents per GPU, P = len(layers_idx) + 1 = 6 image tokens):
        # the same loop at that shape, separately labelled; the 8-rank run is `--gpus 8 --batch 4 --latent 96 --ip-tokens 6`
        set_visual_cross_attention_adapter(unet, (6,))
        unet.to(dev)
        b4, s4, p4 = 4, 96, 6
        k4 = min(args.steps, 15)
        loop4 = DenoiseLoop(unet, b4, s4, p4, max(k4, 3), args.guidance, use_graph=not args.no_graph, share_prefix=False)
        loop4.set_conditioning((mk(b4, 77, 768), mk(b4, p4, 768)), (mk(b4, 77, 768), mk(b4, p4, 768)))
        n4 = torch.randn(b4, 4, s4, s4, generator=g5)
        loop4.reset(n4)
        for _ in range(3):
            loop4.step()
        torch.cuda.synchronize()
        loop4.reset(n4)
        t4 = time.perf_counter()
        for _ in range(k4):
            loop4.step()
        torch.cuda.synchronize()
        dt4 = (time.perf_counter() - t4) / k4
        tf4 = 2 * b4 * 2.1504                        # 2.1504 TFLOP per UNet forward and sample at 96 x 96, P = 6
        cfg4 = {"workload": "configs[4], one rank's share: bs=4 per GPU (32 over 8 ranks), 96x96 latents (768x768), P = 6 image tokens, guidance %.1f" % args.guidance,
                "value": round(1.0 / dt4, 3), "unit": "denoising steps/s (bs=4 per GPU)", "ms_per_step": round(dt4 * 1e3, 3), "steps": k4,
                "step_mfma_frac": round(tf4 / dt4 / MFMA_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(loop4.latents).all().item()),
                "launches_per_step": loop4.launches_per_step}
        loop4 = None
        torch.cuda.empty_cache()
        set_visual_cross_attention_adapter(unet, (5,))
        unet.to(dev)

    # third, separately labelled config (BASELINE configs[3] without the optional ArcFace term): a WHOLE training iteration -
    # adapters + text encoder + UNet forward, the backward through all of them, per-module gradient clipping and AdamW - as two
    # replayed launch plans (photoverse_amd/train.py TrainStep).  NOT part of `value`.
    train_step = None
    if rank == 0 and world == 1 and not args.no_train_forward and S == 64:
        eng = gr = None
        torch.cuda.empty_cache()
        train_step = train_step_leg(unet, B, S, dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (contract); N > 1 runs stay short
        cpu = cpu_baseline()

    if rank == 0:
        workload = workload_label(B, S, P, args.guidance, args.steps, world)
        value = world * args.steps / dt
        step_tflop = 2 * B * UNET_TFLOP_PER_SAMPLE_64 * (S / 64) ** 2 if S == 64 else None
        out = {
            "metric": "denoising steps/sec at 512x512 bs=16, 50-step loop (2 UNet fwd + CFG + DPM-Solver++ step per step)",
            "value": round(value, 3), "unit": "denoising steps/s (bs=16 per GPU, summed over GPUs)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (random-init SD-v1.5-shaped weights, N(0,1) latents / text / image-token embeddings)",
            "config": {"workload": workload,
                       "per_gpu_batch": B, "global_batch": GB, "latent": S, "ip_tokens": P, "guidance_scale": args.guidance,
                       "parallelism": f"dp{world} (batch-sharded, 1 all_gather)", "hip_graph": not args.no_graph, "graph_branches": 1 if args.one_stream else 2 * args.batch_splits,
                       "launches_per_step": launches_per_step},
            "finite": finite, "rccl_world": rccl_world, "collective": ("all_gather_into_tensor over RCCL (final latents)" if use_dist else "none (single process)"),
            "ms_per_step_ranks": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "n": len(rank_ms)},
            "step_mfma_frac": (round(step_tflop / (dt / args.steps) / 1e0 / MFMA_PEAK_TFLOPS, 4) if step_tflop else None),
            "roofline": roofline, "xattn_fused": xfused, "shared_prefix": shared, "ip_tokens_5": ip5, "configs4_per_rank": cfg4, "train_forward": train_fwd, "train_step": train_step, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()               # rank 0's roofline replay is over: every rank leaves the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
