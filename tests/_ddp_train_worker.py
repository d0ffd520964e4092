"""Worker of ``test_data_parallel_training_matches_full_batch``: rank RANK of WORLD_SIZE, all on cuda:0, gloo rendezvous on 127.0.0.1.
Steps its slice of a seeded global batch, all-reduces the gradients with ``GradientReducer``, takes one AdamW step and saves what it holds."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(seed):
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.modeling_utils import load_models
    from oracle.unet_ref import TINY_CONFIG                  # a config dict only: test infrastructure
    VIS = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3, image_size=56, patch_size=14)
    TXT = dict(hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=2)
    models = load_models(None, 1, use_lora=True, lora_config=LoraConfig(r=4, lora_alpha=4, lora_dropout=0.0), unet_config=TINY_CONFIG,
                         vision_config=VIS, text_config=TXT, vae_config=dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1), seed=seed)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = models
    for m in (unet, text_encoder, image_adapter, text_adapter):
        m.to("cuda")
    return unet, text_encoder, text_adapter, image_adapter


def global_batch(G, E, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(noisy_latents=torch.randn(G, 4, 16, 16, generator=g), noise=torch.randn(G, 4, 16, 16, generator=g),
                timesteps=torch.randint(0, 1000, (G,), generator=g), text_input_ids=torch.randint(0, 1000, (G, 77), generator=g),
                placeholder_idx=torch.randint(1, 70, (G, 1), generator=g), image_embeddings=[torch.randn(G, 17, 256, generator=g).half() for _ in range(E)])


def rows(batch, lo, hi):
    out = {}
    for k, v in batch.items():
        if k == "image_embeddings":
            out[k] = [x[lo:hi].cuda() for x in v]
        elif k == "timesteps":
            out[k] = v[lo:hi]
        else:
            out[k] = v[lo:hi].cuda()
    return out


def one_step(models, batch, B, reducer_factory):
    from photoverse_amd.optim import AdamW
    from photoverse_amd.train import TrainStep
    unet, text_encoder, text_adapter, image_adapter = models
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=2, clip_tokens=17, clip_dim=256, grad_scale=512.0)
    groups = step.trainable_parameters()
    params = [p for ps in groups.values() for p in ps]
    opt = AdamW(params, lr=1e-3)
    reducer = reducer_factory(params)
    out = step.step(**batch, forced_fusion=[0.5, 0.1, 0.9, 0.5])
    world = reducer() if reducer is not None else 1
    torch.cuda.synchronize()
    grads = [p.grad.detach().clone().cpu() / (step.grad_scale * world) for p in params]
    opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=step.grad_scale * world)
    torch.cuda.synchronize()
    return dict(loss=out["loss"].cpu(), grads=grads, params=[p.detach().clone().cpu() for p in params], world=world)


def main():
    from photoverse_amd.train import GradientReducer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir, B = sys.argv[1], int(sys.argv[2])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    models = build(101)
    batch = rows(global_batch(B * world, 2, 102), rank * B, (rank + 1) * B)
    res = one_step(models, batch, B, lambda params: GradientReducer(params))
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
