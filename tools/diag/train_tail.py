"""Where the host-bound tail of a training iteration goes (full size, no face branch)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from photoverse_amd.adapters import PhotoVerseAdapter
from photoverse_amd.clip import CLIPTextModel
from photoverse_amd.lora import LoraConfig, inject_adapter_in_model
from photoverse_amd.optim import AdamW
from photoverse_amd.train import TrainStep
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
dev = torch.device("cuda")
torch.manual_seed(0)
unet = UNet2DConditionModel(); set_visual_cross_attention_adapter(unet, (5,)); inject_adapter_in_model(LoraConfig(r=8, lora_alpha=1, lora_dropout=0.1), unet); unet.to(dev)
te, ta, ia = CLIPTextModel().to(dev), PhotoVerseAdapter(1024, 768, 5).to(dev), PhotoVerseAdapter(1024, 768, 5).to(dev)
B, S = 16, 64
ts = TrainStep(unet, te, ta, ia, batch=B, h=S, w=S, n_tokens=5)
groups = ts.trainable_parameters(); opt = AdamW([p for g in groups.values() for p in g], lr=1e-5)
g = torch.Generator().manual_seed(1)
inp = dict(noisy_latents=torch.randn(B, 4, S, S, generator=g).to(dev), noise=torch.randn(B, 4, S, S, generator=g).to(dev), timesteps=torch.randint(0, 1000, (B,), generator=g),
           text_input_ids=torch.randint(0, 49000, (B, 77), generator=g).to(dev), placeholder_idx=torch.full((B, 1), 5).to(dev),
           image_embeddings=[torch.randn(B, 257, 1024, generator=g).half().to(dev) for _ in range(5)])
for _ in range(3):
    ts.step(**inp); opt.step(clip_groups=list(groups.values()), grad_scale=ts.grad_scale)
torch.cuda.synchronize()
def timed(fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("load_weights      %.2f ms" % timed(ts.tape.load_weights))
print("graph replay      %.2f ms" % timed(ts.graph.replay))
print("whole step()      %.2f ms" % timed(lambda: ts.step(**inp)))
print("optimizer.step    %.2f ms" % timed(lambda: opt.step(clip_groups=list(groups.values()), grad_scale=ts.grad_scale)))
print("refresh entries:", len(ts.tape.refresh), " pgrads:", len(ts.pgrads))
