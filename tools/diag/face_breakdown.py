"""Per-launch times of the identity-loss branch's plans (last denoising step + VAE decode + ArcFace, and their backward): every recorded launch replayed
between two events, grouped by (kernel tag, algorithmic flops).  usage (GPU box): python tools/diag/face_breakdown.py"""
import sys, torch
from collections import defaultdict
from types import SimpleNamespace
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from photoverse_amd.adapters import PhotoVerseAdapter
from photoverse_amd.clip import CLIPTextModel
from photoverse_amd.lora import LoraConfig, inject_adapter_in_model
from photoverse_amd.loss import FaceLoss
from photoverse_amd.scheduler import DPMSolverMultistepScheduler
from photoverse_amd.train import TrainStep
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
from photoverse_amd.vae import AutoencoderKL
dev = torch.device("cuda")
torch.manual_seed(0)
unet = UNet2DConditionModel(); set_visual_cross_attention_adapter(unet, (5,)); inject_adapter_in_model(LoraConfig(r=8, lora_alpha=1, lora_dropout=0.1), unet); unet.to(dev)
te, ta, ia = CLIPTextModel().to(dev), PhotoVerseAdapter(1024, 768, 5).to(dev), PhotoVerseAdapter(1024, 768, 5).to(dev)
B, S, ns = 16, 64, 4
ts = TrainStep(unet, te, ta, ia, batch=B, h=S, w=S, n_tokens=5, use_graph=False, grad_scale=4096.0, fusion_seed=1, face_loss=FaceLoss(dev, "arcface"),
               vae=AutoencoderKL().to(dev), noise_scheduler=SimpleNamespace(config=DPMSolverMultistepScheduler().config), face_samples=ns, guidance_scale=2.0, infer_steps=10)
g = torch.Generator().manual_seed(1)
inp = dict(noisy_latents=torch.randn(B, 4, S, S, generator=g).to(dev), noise=torch.randn(B, 4, S, S, generator=g).to(dev), timesteps=torch.randint(0, 1000, (B,), generator=g),
           text_input_ids=torch.randint(0, 49000, (B, 77), generator=g).to(dev), placeholder_idx=torch.full((B, 1), 5).to(dev),
           image_embeddings=[torch.randn(B, 257, 1024, generator=g).half().to(dev) for _ in range(5)])
fi = dict(pixel_values=(torch.rand(ns, 3, 8 * S, 8 * S, generator=g) * 2 - 1).to(dev), start_latents=torch.randn(ns, 4, S, S, generator=g).to(dev),
          image_embeddings=torch.randn(ns, 257, 1024, generator=g).half().to(dev), uncond_image_embeddings=torch.randn(ns, 257, 1024, generator=g).half().to(dev),
          text_input_ids=torch.randint(0, 49000, (ns, 77), generator=g).to(dev), placeholder_idx=torch.full((ns, 1), 4).to(dev),
          uncond_input_ids=torch.randint(0, 49000, (ns, 77), generator=g).to(dev))
for _ in range(2):
    ts.step(**inp, face_inputs=fi)
torch.cuda.synchronize()
s = torch.cuda.current_stream(dev).cuda_stream
fb = ts.face
for label, rec in (("LAST STEP + DECODE + LOSS", fb.rec_last), ("BACKWARD", fb.tape.rb), ("NO-GRAD STEP", fb.loop_tape.rf)):
    fb.state.copy_(torch.tensor([fb.T - 1, fb.T, 0, 0], dtype=torch.int32))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(rec.calls) + 1)]
    for rep in range(2):
        fb.state.copy_(torch.tensor([0 if label.startswith("NO") else fb.T - 1, fb.T, 0, 0], dtype=torch.int32))
        ev[0].record()
        for i, (fn, args) in enumerate(rec.calls):
            fn(*args, s)
            ev[i + 1].record()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for i, tag in enumerate(rec.tags):
        a = agg[(tag[0], tag[1])]
        a[0] += 1; a[1] += ev[i].elapsed_time(ev[i + 1]); a[2] += tag[1]
    tot = sum(v[1] for v in agg.values())
    print(f"{label}: {len(rec.calls)} launches, {tot:.2f} ms (event-serialised)")
    byname = defaultdict(float)
    for (name, fl), (n, ms, f) in agg.items():
        byname[name.split("<")[0]] += ms
    print("  by entry: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:12]))
    for (name, fl), (n, ms, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {ms:7.3f} ms  x{n:3d}  avg {ms / n * 1e3:8.1f} us  {f / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s  {name}  [{fl / 1e9:.1f} GFLOP]")
