"""Generates ``tests/golden/*.pt``.  Run in the BUILD container only
(``python -m oracle.make_golden``): part (1) imports the real reference module
``/root/reference/models/adapters.py`` - the only reference module that imports
without diffusers - which cannot travel to the GPU box.

Fixtures are data only (seeded inputs, weights, expected outputs):
  adapter_golden.pt   real-reference PhotoVerseAdapter (num_tokens=2, default init
                      under seed 1234): per-tensor weight checksums, fp16 inputs,
                      outputs for token_index in {None, 0, 1, 'full'}.
  inject_golden.pt    the worked example of clip.py:21-23 evaluated by hand
                      (idx=5, 5 concept tokens) plus edge rows.
  tiny_unet_golden.pt oracle's own TINY_CONFIG UNet, 2-step denoise (regression
                      fixture for the unpinned restatement; seeds only + outputs).
"""
import os
import sys

import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def adapter_golden():
    sys.path.insert(0, "/root/reference")
    from models.adapters import PhotoVerseAdapter  # the REAL reference class
    torch.manual_seed(1234)
    ad = PhotoVerseAdapter(clip_embedding_dim=1024, cross_attention_dim=768, num_tokens=2).eval()
    # Weights are NOT stored (25 MB): they are the reference's default init under seed 1234.  Per-tensor checksums
    # let the test prove that the restatement built under the same seed holds the same weights.
    sums = {k: (v.double().sum().item(), (v.double() ** 2).sum().item(), v.flatten()[:4].clone()) for k, v in ad.state_dict().items()}
    g = torch.Generator().manual_seed(99)
    embs = [torch.randn(1, 257, 1024, generator=g).half() for _ in range(2)]
    f = [e.float() for e in embs]
    with torch.no_grad():
        outs = {"none": ad(f), "full": ad(f, token_index="full"), "0": ad(f, token_index=0), "1": ad(f, token_index=1)}
    torch.save({"weights_seed": 1234, "weight_checksums": sums, "embs": embs, "outs": outs},
               os.path.join(OUT, "adapter_golden.pt"))
    sys.path.pop(0)


def inject_golden():
    # clip.py:21-23 worked example: seq 77, 5 concept tokens at idx 5:
    #   new[10:] = old[6:73]; new[5:10] = concept; new[:5] = old[:5]
    g = torch.Generator().manual_seed(7)
    old = torch.randn(3, 77, 8, generator=g)
    concept = torch.randn(3, 5, 8, generator=g)
    idx = torch.tensor([[5], [1], [71]])
    exp = old.clone()
    for b, i in enumerate([5, 1, 71]):
        left = 77 - 5 - i
        exp[b, i + 5:] = old[b, i + 1:i + 1 + left]
        exp[b, i:i + 5] = concept[b]
    # hand-written form of the comment, row 0
    assert torch.equal(exp[0, 10:], old[0, 6:73]) and torch.equal(exp[0, 5:10], concept[0]) and torch.equal(exp[0, :5], old[0, :5])
    one = torch.randn(3, 1, 8, generator=g)   # inference default: a single concept token (token_index=0)
    exp1 = old.clone()
    for b, i in enumerate([5, 1, 71]):
        exp1[b, i] = one[b, 0]
    torch.save({"old": old, "concept": concept, "idx": idx, "expected": exp, "one": one, "expected_one": exp1},
               os.path.join(OUT, "inject_golden.pt"))


def tiny_unet_golden():
    from oracle.unet_ref import UNet2DConditionModelRef, TINY_CONFIG, set_visual_cross_attention_adapter_ref
    from oracle.infer_ref import denoise_ref, draw_noise_ref
    torch.manual_seed(0)
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(unet, (5,))
    g = torch.Generator().manual_seed(11)
    text, utext = torch.randn(2, 77, 768, generator=g), torch.randn(2, 77, 768, generator=g)
    ip, uip = torch.randn(2, 1, 768, generator=g), torch.randn(2, 1, 768, generator=g)
    noise = draw_noise_ref(2, 4, 16, seed=3)
    with torch.no_grad():
        eps = unet(noise, torch.tensor(500), encoder_hidden_states=(text, ip)).sample
    lat = denoise_ref(unet, noise, (text, ip), (utext, uip), guidance_scale=7.5, timesteps=2)
    torch.save({"weights_seed": 0, "cond_seed": 11, "noise_seed": 3, "eps_t500": eps, "latents_2step": lat},
               os.path.join(OUT, "tiny_unet_golden.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    adapter_golden()
    inject_golden()
    tiny_unet_golden()
    print("golden fixtures written to", OUT)
