#!/usr/bin/env python3
"""Training CLI: the counterpart of ``/root/reference/train.py`` on the MI355X-native package.

Same flags and defaults for everything the hot path consumes (``train.py:37-285`` of the reference): model / checkpoint paths, AdamW
hyper-parameters, LoRA (``--use_lora --lora_rank --lora_alpha --lora_dropout``), ``--extra_num_tokens`` / ``--image_encoder_layers_idx``,
``--face_loss arcface --face_loss_sample_ratio``, ``--guidance_scale``, ``--lr_scheduler`` / ``--lr_warmup_steps``,
``--data_root_path`` / ``--img_subfolder`` / ``--mask_subfolder`` (the image-folder datasets of ``datasets/custom.py``, with and without masks),
``--checkpoint_save_steps``, ``--samples_save_steps`` / ``--denoise_timesteps`` / ``--num_of_samples_to_save`` (sample grids through
``run_inference``, ``train.py:549-596``), ``--max_train_steps``.  One iteration = ``photoverse_amd.train.training_iteration`` (``train.py:464-549``):
forward + backward launch plans on the HIP kernels, per-module gradient clipping and AdamW on the device; checkpoints in the reference's
``photoverse_XXXXXX.pt`` layout (``models/modeling_utils.py:29-50``), loadable by both code bases.

Build-side additions: ``--pretrained_model_name_or_path random`` (seeded random-init weights, there is no network), ``--tiny`` (small model
sizes for a smoke run), ``--synthetic_data`` (random images instead of a dataset), ``--arcface_weights`` (local state dict of the ArcFace
network; the reference downloads it), ``--image_encoder_path``, ``--grad_scale``.

Not supported (rejected with a message, not ignored): ``--face_loss facenet``
([EXT] facenet_pytorch), ``--report_to`` / ``--push_to_hub`` (no network).  Data-parallel: launched as
``python -m torch.distributed.run --nproc-per-node N train.py ...`` (the reference: ``accelerate launch``) every rank trains its own batches of
``--train_batch_size`` on its own GPU, the trainable gradients are summed by one RCCL all-reduce per optimizer step
(``photoverse_amd.train.GradientReducer``) and rank 0 logs and writes checkpoints.  ``--mixed_precision`` is accepted and ignored: activations are
fp16-stored with fp32 accumulation and fp32 master weights always.  An incomplete last batch of an epoch is dropped (the plans have a
fixed batch size).
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

TEMPLATES = ["a photo of a {}", "a rendering of a {}", "a cropped photo of the {}", "the photo of a {}", "a photo of a clean {}",
             "a photo of a dirty {}", "a dark photo of the {}", "a photo of my {}", "a photo of the cool {}", "a close-up photo of a {}",
             "a bright photo of the {}", "a cropped photo of a {}", "a photo of the {}", "a good photo of the {}", "a photo of one {}",
             "a close-up photo of the {}", "a rendition of the {}", "a photo of the clean {}", "a rendition of a {}", "a photo of a nice {}",
             "a good photo of a {}", "a photo of the nice {}", "a photo of the small {}", "a photo of the weird {}", "a photo of the large {}",
             "a photo of a cool {}", "a photo of a small {}"]                         # datasets/custom.py:10-38 (textual-inversion templates)


def parse_args():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--pretrained_model_name_or_path", type=str, default="random")
    p.add_argument("--pretrained_photoverse_path", type=str, default=None)
    p.add_argument("--image_encoder_path", type=str, default=None)
    p.add_argument("--data_root_path", type=str, default=None)
    p.add_argument("--img_subfolder", type=str, default="images")
    p.add_argument("--mask_subfolder", type=str, default=None)
    p.add_argument("--output_dir", type=str, default="results")
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--lr_warmup_steps", type=int, default=500)
    p.add_argument("--lr_scheduler", type=str, default="constant", choices=["constant", "constant_with_warmup", "linear", "cosine"])
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--adam_weight_decay", type=float, default=1e-2)
    p.add_argument("--adam_epsilon", type=float, default=1e-8)
    p.add_argument("--num_train_epochs", type=int, default=100)
    p.add_argument("--max_train_steps", type=int, default=5000)
    p.add_argument("--train_batch_size", type=int, default=4)
    p.add_argument("--dataloader_num_workers", type=int, default=4)
    p.add_argument("--checkpoint_save_steps", type=int, default=2000)
    p.add_argument("--samples_save_steps", type=int, default=500, help="write a sample grid (input | condition | generated) every N steps; 0 = never")
    p.add_argument("--denoise_timesteps", type=int, default=25)
    p.add_argument("--num_of_samples_to_save", type=int, default=5)
    p.add_argument("--mixed_precision", type=str, default=None)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--extra_num_tokens", type=int, default=4)
    p.add_argument("--image_encoder_layers_idx", type=int, nargs="+", default=[4, 8, 12, 16])
    p.add_argument("--guidance_scale", type=float, default=2.0)
    p.add_argument("--use_random_prompts", action="store_true")
    p.add_argument("--face_loss", type=str, default=None, choices=[None, "arcface", "facenet"])
    p.add_argument("--face_loss_sample_ratio", type=float, default=0.25)
    p.add_argument("--use_lora", action="store_true")
    p.add_argument("--lora_alpha", type=float, default=1)
    p.add_argument("--lora_dropout", type=float, default=0.1)
    p.add_argument("--lora_rank", type=int, default=8)
    p.add_argument("--report_to", type=str, default=None)
    p.add_argument("--push_to_hub", action="store_true")
    p.add_argument("--seed", type=int, default=None)
    # build-side
    p.add_argument("--tiny", action="store_true", help="small random-init model sizes (smoke runs)")
    p.add_argument("--synthetic_data", action="store_true", help="random images instead of --data_root_path")
    p.add_argument("--arcface_weights", type=str, default=None, help="local state dict of the ArcFace IR-ResNet18")
    p.add_argument("--grad_scale", type=float, default=4096.0, help="static loss scale of the fp16 gradient storage")
    p.add_argument("--max_consecutive_skipped_steps", type=int, default=8,
                   help="abort when this many optimizer steps in a row were skipped for non-finite gradients (the loss scale is static: "
                        "a persistent overflow would otherwise train nothing while the LR schedule and the checkpoints advance)")
    args = p.parse_args()
    if len(args.image_encoder_layers_idx) != args.extra_num_tokens:           # train.py:291-292
        p.error("--image_encoder_layers_idx must have --extra_num_tokens entries")
    if args.face_loss == "facenet":
        p.error("--face_loss facenet needs [EXT] facenet_pytorch; only arcface is built")
    if args.gradient_accumulation_steps < 1:
        p.error("--gradient_accumulation_steps must be >= 1")
    if args.report_to or args.push_to_hub:
        p.error("--report_to / --push_to_hub need network access")
    if not args.synthetic_data and not args.data_root_path:
        p.error("give --data_root_path (a folder with <img_subfolder>/<number>.jpg|png) or --synthetic_data")
    return args


def lr_lambda(name, warmup, total):
    """diffusers.optimization.get_scheduler (train.py:380-385): multiplier of the base learning rate at optimizer step ``s``."""
    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: min(1.0, s / max(1, warmup))
    if name == "linear":
        return lambda s: s / max(1, warmup) if s < warmup else max(0.0, (total - s) / max(1, total - warmup))
    return lambda s: s / max(1, warmup) if s < warmup else max(0.0, 0.5 * (1.0 + math.cos(math.pi * (s - warmup) / max(1, total - warmup))))


class ImageFolderDataset(torch.utils.data.Dataset):
    """datasets/custom.py:44-95 (``CustomDataset``): numbered images, the prompt of ``prepare_prompt`` (datasets/utils.py:160-220)."""

    def __init__(self, data_root, tokenizer, img_subfolder="images", size=512, placeholder_token="*", template="a photo of {}",
                 use_random_templates=False):
        img_dir = os.path.join(data_root, img_subfolder)
        paths = [os.path.join(img_dir, f) for f in os.listdir(img_dir) if any(e in f.lower() for e in ("jpg", "png", "jpeg"))]
        self.image_paths = sorted(paths, key=lambda x: int(os.path.basename(x).split(".")[0]))
        self.tokenizer, self.size, self.placeholder_token, self.template = tokenizer, size, placeholder_token, template
        self.use_random_templates = use_random_templates

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        import numpy as np
        from PIL import Image
        from photoverse_amd.image_utils import clip_image_processor, preprocess_image
        template = TEMPLATES[np.random.randint(len(TEMPLATES))] if self.use_random_templates else self.template
        ex = prompt_example(self.tokenizer, template, self.placeholder_token)
        raw = Image.open(self.image_paths[idx])
        if raw.mode != "RGB":
            raw = raw.convert("RGB")
        ex["pixel_values"] = preprocess_image(raw, size=self.size, interpolation="bicubic")
        ex["pixel_values_clip"] = clip_image_processor(raw)
        return ex


def crop_box_of_mask(mask, grow=0.15):
    """The crop ``CustomDatasetWithMasks._crop_to_mask_and_scale`` takes (datasets/custom.py:143-171): bounding box of the non-zero mask,
    each side moved out by ``grow`` x the box size (clipped to the image), then the SHORTER side widened by half of the longer side's
    length on both of its ends (clipped) - the reference's way of getting a roughly square, generously padded crop.  (ymin, ymax, xmin, xmax)."""
    import numpy as np
    ys, xs = np.where(np.asarray(mask) > 0)
    H, W = np.asarray(mask).shape[:2]
    ymin, ymax, xmin, xmax = int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())
    bh, bw = ymax - ymin, xmax - xmin
    ymin, ymax = max(0, int(ymin - bh * grow)), min(H, int(ymax + bh * grow))
    xmin, xmax = max(0, int(xmin - bw * grow)), min(W, int(xmax + bw * grow))
    cw, ch = xmax - xmin, ymax - ymin
    if cw > ch:
        ymax, ymin = min(H, ymax + cw // 2), max(0, ymin - cw // 2)
    elif ch > cw:
        xmax, xmin = min(W, xmax + ch // 2), max(0, xmin - ch // 2)
    return ymin, ymax, xmin, xmax


class MaskedImageFolderDataset(ImageFolderDataset):
    """datasets/custom.py:97-141 (``CustomDatasetWithMasks``): the CLIP-side image is the photo resized to the mask's size, blacked out
    outside the mask and cropped around it; the VAE-side pixels are the full photo."""

    def __init__(self, data_root, tokenizer, img_subfolder="images", mask_subfolder="masks", **kw):
        super().__init__(data_root, tokenizer, img_subfolder, **kw)
        mask_dir = os.path.join(data_root, mask_subfolder)
        paths = [os.path.join(mask_dir, f) for f in os.listdir(mask_dir) if any(e in f.lower() for e in ("jpg", "png", "jpeg"))]
        self.mask_paths = sorted(paths, key=lambda x: int(os.path.basename(x).split(".")[0]))
        if len(self.mask_paths) != len(self.image_paths):
            raise ValueError(f"{len(self.image_paths)} images but {len(self.mask_paths)} masks under {data_root}")

    def __getitem__(self, idx):
        import numpy as np
        from PIL import Image
        from photoverse_amd.image_utils import clip_image_processor, preprocess_image
        template = TEMPLATES[np.random.randint(len(TEMPLATES))] if self.use_random_templates else self.template
        ex = prompt_example(self.tokenizer, template, self.placeholder_token)
        raw = Image.open(self.image_paths[idx])
        mask = Image.open(self.mask_paths[idx])
        raw = raw if raw.mode == "RGB" else raw.convert("RGB")
        mask = mask if mask.mode == "L" else mask.convert("L")
        small = np.array(raw.resize(mask.size))
        m = np.array(mask)
        masked = np.where((m > 0)[:, :, None], small, 0).astype(np.uint8)
        y0, y1, x0, x1 = crop_box_of_mask(m)
        ex["pixel_values"] = preprocess_image(raw, size=self.size, interpolation="bicubic")
        ex["pixel_values_clip"] = clip_image_processor(Image.fromarray(masked[y0:y1, x0:x1]))
        return ex


def prompt_example(tokenizer, template, placeholder_token):
    text = template.format(placeholder_token)
    ids = tokenizer(text, padding="max_length", truncation=True, max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    return {"text_input_ids": ids[0], "concept_placeholder_idx": torch.tensor([text.split().index(placeholder_token) + 1])}


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, tokenizer, size, clip_size, n=64, seed=0):
        self.tokenizer, self.size, self.clip_size, self.n, self.seed = tokenizer, size, clip_size, n, seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 100003 + idx)
        ex = prompt_example(self.tokenizer, "a photo of {}", "*")
        ex["pixel_values"] = torch.rand(3, self.size, self.size, generator=g) * 2 - 1
        ex["pixel_values_clip"] = torch.randn(3, self.clip_size, self.clip_size, generator=g)
        return ex


def collate(examples):                                      # datasets/utils.py collate_fn
    return {k: torch.stack([e[k] for e in examples]) for k in examples[0]}


def save_samples(args, global_step, batch, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, noise_scheduler, device, face):
    """A grid of input / condition / generated images for the current batch with the weights as they are now (train.py:555-596)."""
    from photoverse_amd.image_utils import denormalize, denormalize_clip, save_images_grid, to_pil
    from photoverse_amd.infer import run_inference
    n = min(args.num_of_samples_to_save, batch["pixel_values"].shape[0])
    unet.repack()                                           # the inference engines bake fp16 copies of the weights: rebuild from the trained ones
    example = {k: (v[:n] if torch.is_tensor(v) else v) for k, v in batch.items()}
    if args.use_random_prompts:                             # the grid is generated from the plain prompt, not the batch's random templates (train.py:558-560)
        pe = prompt_example(tokenizer, "a photo of {}", "*")
        example["text_input_ids"] = pe["text_input_ids"].unsqueeze(0).expand(n, -1).contiguous().to(example["text_input_ids"].device)
        example["concept_placeholder_idx"] = pe["concept_placeholder_idx"].unsqueeze(0).expand(n, -1).contiguous().to(example["concept_placeholder_idx"].device)
    with torch.no_grad():
        gen = run_inference(example, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, noise_scheduler, device,
                            args.image_encoder_layers_idx, latent_size=args.resolution // 8, guidance_scale=args.guidance_scale,
                            timesteps=args.denoise_timesteps, token_index=0, disable_tqdm=True)
    if face is not None:
        sim = float(face(example["pixel_values"].to(device, torch.float32), gen.float(), normalize=False, maximize=False))
        print(f"step {global_step}: face_similarity={sim:.4f}", flush=True)
    grid = [("Input Images", [to_pil(denormalize(i)) for i in example["pixel_values"]]),
            ("Condition Images", [to_pil(denormalize_clip(i)).resize((args.resolution, args.resolution)) for i in example["pixel_values_clip"]]),
            ("a photo of {}", [to_pil(denormalize(i)) for i in gen.float().cpu()])]
    save_images_grid(grid, os.path.join(args.output_dir, f"{str(global_step).zfill(5)}.jpg"))


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a HIP device: photoverse_amd has no CPU path")
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.loss import FaceLoss
    from photoverse_amd.modeling_utils import load_models, save_progress
    from photoverse_amd.optim import AdamW
    from photoverse_amd.train import GradientReducer, TrainStep, training_iteration
    # data-parallel: one process per GPU under ``python -m torch.distributed.run --nproc-per-node N train.py ...`` (the reference: ``accelerate launch``)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    device = torch.device("cuda", local if world > 1 else torch.cuda.current_device())
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=os.environ.get("PV_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    main_process = rank == 0
    if args.seed is not None:
        torch.manual_seed(args.seed)
    cfg, clip_size = {}, 224
    if args.tiny:
        if args.pretrained_model_name_or_path != "random":
            raise SystemExit("--tiny builds a small random-init model: use it with --pretrained_model_name_or_path random")
        clip_size = 56
        cfg = dict(unet_config=dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D")),
                   vision_config=dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=max(args.image_encoder_layers_idx) + 1,
                                      image_size=56, patch_size=14),
                   text_config=dict(hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=2),
                   vae_config=dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1))
    lora_config = LoraConfig(r=args.lora_rank, lora_alpha=args.lora_alpha, lora_dropout=args.lora_dropout) if args.use_lora else None   # train.py:346-354
    tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, noise_scheduler, lora_config = load_models(
        None if args.pretrained_model_name_or_path == "random" else args.pretrained_model_name_or_path, args.extra_num_tokens,
        args.pretrained_photoverse_path, use_lora=args.use_lora, lora_config=lora_config, image_encoder_path=args.image_encoder_path,
        seed=args.seed or 0, **cfg)
    for m in (vae, unet, text_encoder, image_encoder, image_adapter, text_adapter):
        m.to(device)
    face = None
    if args.face_loss:
        # the reference builds ArcFaceResNet18(pretrained=True) (models/loss.py:21-22): an identity loss against a randomly initialised
        # network is meaningless, so weights are REQUIRED unless the whole run is a random-init smoke run
        smoke = args.tiny or args.pretrained_model_name_or_path in (None, "random")
        if not args.arcface_weights and not smoke:
            raise SystemExit("--face_loss arcface needs --arcface_weights <state dict of the ArcFace IR-ResNet18> (the reference downloads "
                             "them, this build has no network); only --tiny / random-init runs may use a randomly initialised ArcFace")
        face = FaceLoss(device, args.face_loss)
        if args.arcface_weights:
            sd = torch.load(args.arcface_weights, map_location="cpu")
            # the published checkpoint is a DataParallel state dict (models/arcface_resnet.py:131-134): strip its 'module.' prefix
            sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
            face.model.load_state_dict(sd)
            face.model.to(device)
        elif main_process:
            print("WARNING: --face_loss arcface without --arcface_weights: the identity loss uses a RANDOMLY INITIALISED ArcFace network "
                  "(smoke run only)", flush=True)
    os.makedirs(args.output_dir, exist_ok=True)
    lat = args.resolution // 8
    B = args.train_batch_size
    vis = image_encoder.config
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=lat, w=lat, n_tokens=args.extra_num_tokens + 1,
                     clip_tokens=(vis.image_size // vis.patch_size) ** 2 + 1, clip_dim=vis.hidden_size, grad_scale=args.grad_scale,
                     fusion_seed=args.seed or 0, face_loss=face, vae=vae if face is not None else None, noise_scheduler=noise_scheduler,
                     face_samples=max(int(args.face_loss_sample_ratio * B), 1), guidance_scale=args.guidance_scale, image_size=args.resolution)
    groups = step.trainable_parameters()
    # parameter ORDER as the reference builds it (train.py:366-372: image_adapter, text_adapter, then the UNet's trainable parameters in
    # named_parameters() order): torch.optim state dicts are keyed by index, so the 'optimizer' entry of a checkpoint is only portable
    # between the two code bases with the same order (the two adapters have identical shapes - a swap would load without an error)
    trainable = {id(p) for g in groups.values() for p in g}
    ordered = list(image_adapter.parameters()) + list(text_adapter.parameters()) + [p for _n, p in unet.named_parameters() if id(p) in trainable]
    assert {id(p) for p in ordered} == trainable and len(ordered) == len(trainable)
    optimizer = AdamW(ordered, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                      weight_decay=args.adam_weight_decay, eps=args.adam_epsilon)                                         # train.py:372-377
    sched = lr_lambda(args.lr_scheduler, args.lr_warmup_steps, args.max_train_steps)
    if args.synthetic_data:
        dataset = SyntheticDataset(tokenizer, args.resolution, clip_size, n=max(B * 4, 16), seed=args.seed or 0)
    else:
        kw = dict(size=args.resolution, use_random_templates=args.use_random_prompts)                                 # train.py:388-396
        dataset = (ImageFolderDataset(args.data_root_path, tokenizer, args.img_subfolder, **kw) if args.mask_subfolder is None else
                   MaskedImageFolderDataset(args.data_root_path, tokenizer, args.img_subfolder, args.mask_subfolder, **kw))
    sampler = None
    if world > 1:                                            # accelerator.prepare(train_dataloader): every rank draws its own batches of B
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=args.seed or 0, drop_last=True)
    loader = torch.utils.data.DataLoader(dataset, shuffle=sampler is None, sampler=sampler, collate_fn=collate, batch_size=B, drop_last=True,
                                         num_workers=0 if args.synthetic_data else args.dataloader_num_workers)
    reducer = GradientReducer([p for g in groups.values() for p in g]) if world > 1 else None
    # noise / timestep draws: one stream per rank (weights and the fusion-draw seed are the same on every rank, the data and the noise are not)
    gen = torch.Generator().manual_seed(args.seed + rank) if args.seed is not None else None
    global_step, micro = 0, 0
    last_skipped, consecutive_skips = 0, 0
    acc_n = args.gradient_accumulation_steps
    for epoch in range(args.num_train_epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch in loader:
            optimizer.lr = args.learning_rate * sched(global_step)
            out = training_iteration(step, optimizer, batch, tokenizer, image_encoder, vae, noise_scheduler, device, args.image_encoder_layers_idx,
                                     args.extra_num_tokens, generator=gen, micro_step=micro, accumulation_steps=acc_n, reducer=reducer)
            micro = (micro + 1) % acc_n
            if micro:                                        # accelerator.accumulate: no optimizer step, no logging yet (train.py:464, :551)
                continue
            global_step += 1
            logs = {"loss_mle": float(out["diffusion_loss"]), "loss_reg_concept_text": float(out["concept_text_loss"]),
                    "loss_reg_cross_attn_visual": float(out["cross_attn_visual_loss"]), "lr": optimizer.lr}                    # train.py:612-617
            if face is not None:
                logs["loss_face"] = float(out["face_loss"])
            if main_process:
                print(f"step {global_step}: " + ", ".join(f"{k}={v:.6g}" for k, v in logs.items()), flush=True)
            # fp16 activations / gradients under a static loss scale can overflow where the fp32 reference cannot.  The optimizer skips steps
            # whose gradient norm is not finite (optim.AdamW; the gradients are all-reduced, so every rank takes the same decision); a
            # non-finite LOSS means the forward itself overflowed.  Losses are rank-local: agree on the flag first, so that every rank leaves
            # together instead of one rank exiting and the others hanging in the next all-reduce until the RCCL timeout.
            bad = not all(math.isfinite(v) for v in logs.values())
            if world > 1:
                import torch.distributed as dist
                flag = torch.tensor([1.0 if bad else 0.0], device="cpu" if dist.get_backend() == "gloo" else device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                bad = bool(flag.item())
            skipped = optimizer.skipped_steps
            consecutive_skips = consecutive_skips + 1 if skipped > last_skipped else 0
            last_skipped = skipped
            if main_process and consecutive_skips:
                print(f"step {global_step}: optimizer step SKIPPED (non-finite gradients under --grad_scale {args.grad_scale:g}); "
                      f"{skipped} skipped so far, {consecutive_skips} in a row", flush=True)
            too_many = consecutive_skips >= args.max_consecutive_skipped_steps
            if bad or too_many:
                if world > 1:
                    dist.destroy_process_group()
                why = (f"non-finite loss at step {global_step} on at least one rank (this rank: {logs})" if bad else
                       f"{consecutive_skips} optimizer steps in a row skipped for non-finite gradients at step {global_step}")
                raise SystemExit(f"{why}; optimizer skipped {skipped} step(s) so far - lower --grad_scale or the learning rate")
            if main_process and args.samples_save_steps and global_step % args.samples_save_steps == 0:                            # train.py:555-596
                save_samples(args, global_step, batch, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae,
                             noise_scheduler, device, face)
            if main_process and global_step % args.checkpoint_save_steps == 0:
                print(f"checkpoint at step {global_step}: {optimizer.applied_steps} optimizer steps applied, {skipped} skipped", flush=True)
                save_progress(image_adapter, text_adapter, unet, None, args.output_dir, step=global_step, lora_config=lora_config, optimizer=optimizer)
            if global_step >= args.max_train_steps:
                break
        if global_step >= args.max_train_steps:
            break
    if main_process:
        save_progress(image_adapter, text_adapter, unet, None, args.output_dir, lora_config=lora_config, optimizer=optimizer)   # train.py:627-629
        print(f"saved {os.path.join(args.output_dir, 'photoverse.pt')} after {global_step} steps")
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
