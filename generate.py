#!/usr/bin/env python3
"""Inference CLI - counterpart of ``/root/reference/generate.py`` for the MI355X build.

Same flags and the same call into ``run_inference`` as the reference (``generate.py:21-34,82-85``); additions forced by
the offline environment: ``--model_path random`` (seeded random-init weights), ``--synthetic_input`` (no image / tokenizer
files needed), ``--seed``, ``--latent_size``.  Images are written as PNG like the reference (``generate.py:86-90``).
"""
import argparse
import os

import torch

from photoverse_amd.infer import run_inference
from photoverse_amd.modeling_utils import load_models

parser = argparse.ArgumentParser(description="Run PhotoVerse inference on MI355X")
parser.add_argument("--model_path", type=str, default="random", help="Local HF-layout model directory, or 'random'")
parser.add_argument("--extra_num_tokens", type=int, default=4, help="Number of additional tokens")
parser.add_argument("--encoder_layers_idx", nargs="+", type=int, default=[4, 8, 12, 16], help="Indices of image encoder layers")
parser.add_argument("--guidance_scale", type=float, default=1.0, help="Guidance scale")
parser.add_argument("--checkpoint_path", type=str, default=None, help="Path to a photoverse*.pt checkpoint")
parser.add_argument("--input_image_path", type=str, default=None, help="Path to the input image (needs PIL)")
parser.add_argument("--output_image_path", type=str, default="generated_image", help="Prefix for the outputs")
parser.add_argument("--num_timesteps", type=int, default=25, help="Number of timesteps for inference")
parser.add_argument("--results_dir", type=str, default="results", help="Directory to save the outputs")
parser.add_argument("--text", type=str, default="a photo of a {}", help="Prompt template")
parser.add_argument("--negative_prompt", type=str, default=None, help="Negative prompt")
parser.add_argument("--num_of_samples", type=int, default=None, help="Number of samples to generate")
parser.add_argument("--from_noised_image", action="store_true", help="Use noised image as input (needs a VAE)")
parser.add_argument("--synthetic_input", action="store_true", help="Random CLIP pixels instead of an image file")
parser.add_argument("--seed", type=int, default=None)
parser.add_argument("--latent_size", type=int, default=64)
parser.add_argument("--image_encoder_path", type=str, default=None,
                    help="Local directory of openai/clip-vit-large-patch14 (default: <model_path>/image_encoder)")
parser.add_argument("--tiny", action="store_true", help="Small random-init model (smoke tests of the CLI; needs --model_path random)")


def prepare_example(args, tokenizer):
    """Output format of ``datasets/utils.py:160-199`` (prepare_prompt) + ``generate.py:53-61``."""
    n = args.num_of_samples or 1
    placeholder = "*"
    text = args.text.format(placeholder)
    ids = tokenizer([text] * n, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    idx = text.split().index(placeholder) + 1                     # + BOS (datasets/utils.py:215-220)
    example = {"text": [text] * n, "text_input_ids": ids, "concept_placeholder_idx": torch.full((n, 1), idx, dtype=torch.int64)}
    if args.negative_prompt is not None:
        example["negative_text_input_ids"] = tokenizer([args.negative_prompt] * n, padding="max_length",
                                                       max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    if args.synthetic_input or args.input_image_path is None:
        g = torch.Generator().manual_seed(0 if args.seed is None else args.seed)
        example["pixel_values_clip"] = torch.randn(1, 3, 224, 224, generator=g).repeat(n, 1, 1, 1)
        example["pixel_values"] = torch.zeros(n, 3, 8 * args.latent_size, 8 * args.latent_size)
    else:
        from PIL import Image
        from photoverse_amd.image_utils import clip_image_processor, preprocess_image
        raw_image = Image.open(args.input_image_path)
        if raw_image.mode != "RGB":
            raw_image = raw_image.convert("RGB")
        # generate.py:57-58: CLIPImageProcessor (short side 224 bicubic + centre crop) and preprocess_image (short side `size`
        # bicubic + centre crop, [-1, 1]); size = 8 * latent (512 for the reference's fixed latent_size 64)
        example["pixel_values_clip"] = clip_image_processor(raw_image)[None].repeat(n, 1, 1, 1)
        example["pixel_values"] = preprocess_image(raw_image, size=8 * args.latent_size, interpolation="bicubic")[None].repeat(n, 1, 1, 1)
    return example


if __name__ == "__main__":
    args = parser.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("generate.py needs a HIP device: photoverse_amd has no CPU path")
    device = torch.device("cuda")
    cfg = {}
    if args.tiny:
        if args.model_path != "random":
            raise SystemExit("--tiny builds a small random-init model: use it with --model_path random")
        cfg = dict(unet_config=dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D")),
                   vision_config=dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=max(args.encoder_layers_idx) + 1),
                   text_config=dict(hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=2),
                   vae_config=dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1))      # 4 levels: x8 like the real VAE
    tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None if args.model_path == "random" else args.model_path, args.extra_num_tokens, args.checkpoint_path,
        image_encoder_path=args.image_encoder_path, **cfg)
    for m in (vae, unet, text_encoder, image_encoder, image_adapter, text_adapter):
        m.to(device)
    example = prepare_example(args, tokenizer)
    with torch.no_grad():
        out = run_inference(example, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, scheduler, device,
                            args.encoder_layers_idx, latent_size=args.latent_size, guidance_scale=args.guidance_scale,
                            timesteps=args.num_timesteps, from_noised_image=args.from_noised_image, seed=args.seed)
    os.makedirs(args.results_dir, exist_ok=True)
    from photoverse_amd.image_utils import denormalize, to_pil
    imgs = [to_pil(denormalize(img)) for img in out.float().cpu()]                            # generate.py:86
    for idx, img in enumerate(imgs):
        img.save(os.path.join(args.results_dir, f"{args.output_image_path}{idx}.png"))
    print(f"saved {len(imgs)} image(s) {tuple(out.shape[1:])} to {args.results_dir}/")
