"""Graph-captured denoising loop (the hot loop of ``/root/reference/models/infer.py:98-119``) and its batch sharding.

One denoising step = UNet(uncond) + UNet(cond) at batch B (two sequential forwards, ``infer.py:103-114``) + CFG combine
(``:116``) + ``scheduler.step`` (``:119``).  All of it is a fixed list of allocation-free launches over static buffers,
so ONE step is captured into a HIP graph and replayed ``T`` times; the per-step scalars (timestep, solver coefficients)
are read on the device through a step counter the graph itself advances.

Multi-GPU (new - the reference has none): samples are independent, so the batch is sharded over ranks
with no communication inside the loop; final latents are collected with ONE ``all_gather`` over RCCL.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from .ops import Recorder, require_cuda
from .scheduler import DPMSolverMultistepScheduler


#: 256-row-tile threshold of plans that run beside their CFG twin (each fills half of the chip); PV_SIDE_BIG_MIN overrides it for A/Bs
_SIDE_BIG_MIN = int(os.environ.get("PV_SIDE_BIG_MIN", "128"))


#: rows (batch 2B x pixels of the first merged level) from which the low-resolution levels of the two CFG forwards run as ONE merged plan
_MERGE_MIN_ROWS = int(os.environ.get("PV_MERGE_MIN_ROWS", "8192"))


class DenoiseLoop:
    def __init__(self, unet, batch: int, latent_size: int, n_ip: int, num_steps: int, guidance_scale: float,
                 scheduler: Optional[DPMSolverMultistepScheduler] = None, n_text: int = 77, use_graph: bool = True,
                 two_streams: bool = True, batch_splits: int = 1, training_mode: bool = False, fusion_seed: int = 0,
                 merge_lowres: Optional[bool] = None, share_prefix: Optional[bool] = None):
        """``training_mode``: the reference enables grad on the LAST denoising step only (infer.py:99), where every cross-attention
        layer of both forwards then draws its branch fusion (attention_processor.py:413-420).  Here the draw runs on the device inside
        the captured step (``pv_fusion_draw`` keyed on the step counter), so the same graph serves all steps.  This is the forward semantics
        of that mode; the differentiated last step lives in ``train.TrainStep(face_loss=...)``.

        ``merge_lowres`` (default: on from 8192 rows at the first merged level; env ``PV_MERGE_LOWRES`` forces it): the two CFG forwards (infer.py:103-114) run their two highest-resolution levels
        as two parallel graph branches, but everything below (16 x 16 and 8 x 8 levels, mid block) as ONE plan over both branches' samples:
        at M = B * 256 / B * 64 rows a single branch cannot fill the chip without split-K (fp32 slabs + a reduce launch per conv) and both
        branches stream the same 29.5 MB of weights per conv.  Samples never interact inside the UNet, so the result per sample is unchanged.

        ``share_prefix`` (default: env ``PV_SHARE_PREFIX``, off - the headline metric counts two FULL forwards per step): conv_in, the first ResnetBlock and the first
        transformer block up to its self-attention never see the conditioning, so the uncond and cond forwards of a step compute them twice on
        identical inputs.  With ``share_prefix`` they are one plan whose outputs both branches start from - bit-identical latents, 2.5 % fewer
        flops per step.  Reported by ``bench.py`` as a separately labelled number."""
        dev = unet.device
        if dev.type != "cuda":
            raise RuntimeError("DenoiseLoop needs the UNet on a HIP device (no CPU path)")
        self.unet, self.B, self.S, self.P, self.T = unet, batch, latent_size, n_ip, num_steps
        self.guidance = float(guidance_scale)
        self.training_mode = bool(training_mode)
        sch = scheduler if scheduler is not None else DPMSolverMultistepScheduler()
        sch.set_timesteps(num_steps)
        self.scheduler = sch
        cfg = unet.config
        xdim = cfg.cross_attention_dim
        f32, f16 = torch.float32, torch.float16
        self.latents = torch.zeros((batch, cfg.in_channels, latent_size, latent_size), dtype=f32, device=dev)
        self.x0_prev = torch.zeros_like(self.latents)
        self.timesteps = sch.timesteps.to(device=dev, dtype=f32)
        self.coef = sch.coefficient_table().to(dev)
        self.state = torch.tensor([0, num_steps, 0, 0], dtype=torch.int32, device=dev)   # {step index, table rows, -, -}
        self._state0 = self.state.clone()
        self._host_step = 0
        self.text_c = torch.zeros((batch * n_text, xdim), dtype=f16, device=dev)
        self.text_u = torch.zeros_like(self.text_c)
        self.ip_c = torch.zeros((batch * n_ip, xdim), dtype=f16, device=dev)
        self.ip_u = torch.zeros_like(self.ip_c)
        self.two_streams = two_streams
        # Engines: one per (CFG branch, sub-batch).  batch_splits > 1 cuts each forward into independent sub-batches (samples
        # never interact inside the UNet) that run as additional parallel graph branches.
        if batch % batch_splits:
            raise ValueError("batch_splits must divide the batch")
        sb = batch // batch_splits
        self.eps_u = torch.empty_like(self.latents)
        self.eps_c = torch.empty_like(self.latents)
        self.engines_u, self.engines_c, self.engines_m, self.engines_p = [], [], [], []
        if share_prefix is None:
            share_prefix = os.environ.get("PV_SHARE_PREFIX", "0") == "1"
        first = unet.down_blocks[0]
        self.share_prefix = bool(share_prefix and not training_mode and batch_splits == 1 and getattr(first, "has_attn", False) and len(first.resnets) >= 1)
        pre_kw = {}
        if self.share_prefix:
            self.engines_p.append(unet.engine(batch, latent_size, latent_size, n_ip, 1, latents_in=self.latents, segment="prefix", timesteps=self.timesteps,
                                              state=self.state, n_text=n_text))
            pre_kw = dict(prefix=self.engines_p[0].prefix_out)
        n_lv = len(cfg.block_out_channels)
        split = int(os.environ.get("PV_MERGE_SPLIT", "2"))     # resolution levels that stay in the per-branch plans (A/B switch; 3 = only 8 x 8 + mid merged)
        if merge_lowres is None:
            # default: merge where the merged plan, which runs ALONE on the chip, can fill it - from 8192 rows at its first level (batch 2B) on: the headline
            # (2 x 16 x 256 rows) merges (+ round 4); configs[4]'s per-rank shape (2 x 4 x 576 = 4608 rows) is 4.8 % of a step faster with its two
            # low-resolution plans side by side on the two streams (round 6, profiles/r06_loop_ab_cfg4_env.txt).  PV_MERGE_LOWRES=1 / 0 forces it.
            env = os.environ.get("PV_MERGE_LOWRES")
            rows = 2 * batch * (latent_size >> split) ** 2
            # (... and at the launch-bound end, up to 1024 rows - bs = 1 / 2 at 64 x 64 latents - the merged plan's fewer launches win again: +1.0 % / +0.4 %)
            merge_lowres = (env != "0") if env is not None else (rows >= _MERGE_MIN_ROWS or rows <= 1024)
        self.merge_lowres = bool(merge_lowres and not training_mode and batch_splits == 1 and n_lv > split
                                 and ((latent_size >> split) ** 2) % 64 == 0 and latent_size % (1 << (n_lv - 1)) == 0)
        if self.merge_lowres:
            # text / image-token buffers of the two branches are the halves of ONE buffer: the merged plan reads it whole
            text_all = torch.zeros((2 * batch * n_text, xdim), dtype=f16, device=dev)
            ip_all = torch.zeros((2 * batch * n_ip, xdim), dtype=f16, device=dev)
            self.text_u, self.text_c = text_all[:batch * n_text], text_all[batch * n_text:]
            self.ip_u, self.ip_c = ip_all[:batch * n_ip], ip_all[batch * n_ip:]
            boc = cfg.block_out_channels
            n_in, n_out = (latent_size >> split) ** 2, (latent_size >> (split - 1)) ** 2       # pixels per sample at the two seams
            c_in, c_out = boc[split - 1], boc[split]           # downsampler of level split-1 keeps its width; upsampler of the first merged up block
            mid_in = torch.empty((2 * batch * n_in, c_in), dtype=f16, device=dev)
            mid_in_cs = torch.zeros((2 * batch * n_in // 64, 2, c_in), dtype=f32, device=dev)
            mid_out = torch.empty((2 * batch * n_out, c_out), dtype=f16, device=dev)
            mid_out_cs = torch.zeros((2 * batch * n_out // 64, 2, c_out), dtype=f32, device=dev)
            kw = dict(timesteps=self.timesteps, state=self.state, n_text=n_text, split=split)
            side_by_side = dict(big_min=_SIDE_BIG_MIN) if two_streams else {}      # heads / tails of the two branches run concurrently: half the chip each
            for i, (text, ip, eps, lst) in enumerate(((self.text_u, self.ip_u, self.eps_u, self.engines_u), (self.text_c, self.ip_c, self.eps_c, self.engines_c))):
                half_in = (mid_in[i * batch * n_in:(i + 1) * batch * n_in], mid_in_cs[i * batch * n_in // 64:(i + 1) * batch * n_in // 64])
                half_out = (mid_out[i * batch * n_out:(i + 1) * batch * n_out], mid_out_cs[i * batch * n_out // 64:(i + 1) * batch * n_out // 64])
                lst.append(unet.engine(batch, latent_size, latent_size, n_ip, 1, latents_in=self.latents, text=text, ip=ip, out=eps, segment="outer",
                                       mid_in=half_in, mid_out=half_out, **kw, **side_by_side, **pre_kw))
            self.engines_m.append(unet.engine(2 * batch, latent_size, latent_size, n_ip, 1, text=text_all, ip=ip_all, segment="mid",
                                              mid_in=(mid_in, mid_in_cs), mid_out=(mid_out, mid_out_cs), **kw))
        for i in range(0 if self.merge_lowres else batch_splits):
            sl = slice(i * sb, (i + 1) * sb)
            kw = dict(timesteps=self.timesteps, state=self.state, latents_in=self.latents[sl], n_text=n_text, **pre_kw)
            if two_streams and batch_splits == 1:
                kw.update(big_min=_SIDE_BIG_MIN)                    # the two whole forwards run side by side
            if training_mode:
                kw.update(device_fusion="last_step")
            fs = dict(fusion_seed=fusion_seed * 4096 + 2 * i) if training_mode else {}
            fs2 = dict(fusion_seed=fusion_seed * 4096 + 2 * i + 1) if training_mode else {}
            self.engines_u.append(unet.engine(sb, latent_size, latent_size, n_ip, 1, text=self.text_u[i * sb * n_text:(i + 1) * sb * n_text],
                                              ip=self.ip_u[i * sb * n_ip:(i + 1) * sb * n_ip], out=self.eps_u[sl], **kw, **fs))
            self.engines_c.append(unet.engine(sb, latent_size, latent_size, n_ip, 1, text=self.text_c[i * sb * n_text:(i + 1) * sb * n_text],
                                              ip=self.ip_c[i * sb * n_ip:(i + 1) * sb * n_ip], out=self.eps_c[sl], **kw, **fs2))
        self.eng_u, self.eng_c = self.engines_u[0], self.engines_c[0]
        self.tail = Recorder(dev)
        self.tail.cfg_dpm_step(self.eps_u, self.eps_c, self.latents, self.x0_prev, self.coef, self.state, self.guidance)
        self.tail.step_advance(self.state)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.use_graph = use_graph
        self.two_streams = two_streams
        n_side = (2 * batch_splits - 1) if two_streams else 0
        self._sides = [torch.cuda.Stream(device=dev) for _ in range(n_side)]
        self.launches_per_step = sum(len(e.rec) for e in self.all_engines) + len(self.tail)

    @property
    def all_engines(self):
        """Every plan of a step (uncond / cond branches and, with ``merge_lowres``, the merged low-resolution part)."""
        return self.engines_u + self.engines_c + self.engines_m + self.engines_p

    # ------------------------------------------------------------------
    def set_conditioning(self, cond: Tuple[torch.Tensor, torch.Tensor], uncond: Tuple[torch.Tensor, torch.Tensor]):
        """cond / uncond = (text (B,77,768), ip (B,P,768)) - the tuples of ``infer.py:106,113``."""
        for (text, ip), (dt, di) in ((cond, (self.text_c, self.ip_c)), (uncond, (self.text_u, self.ip_u))):
            require_cuda(text, "text embeddings")
            dt.copy_(text.reshape(dt.shape))
            di.copy_(ip.reshape(di.shape))
        # K/V projections of the conditioning: once per generation, outside the per-step graph
        for e in self.all_engines:
            e.run_conditioning()

    def reset(self, noise: torch.Tensor):
        """latents = noise * init_noise_sigma (``infer.py:70``); step counter to 0."""
        self.latents.copy_(noise.to(self.latents.device) * self.scheduler.init_noise_sigma)
        self.x0_prev.zero_()
        self.state.copy_(self._state0)
        self._host_step = 0

    def _step_eager(self):
        for e in self.engines_p:                       # the conditioning-independent prefix both branches start from (share_prefix)
            e.rec.run()
        if self.merge_lowres:
            # head_u || head_c -> merged low-resolution part (both branches' samples as one batch) -> tail_u || tail_c -> CFG + solver step
            (eu,), (ec,), (em,) = self.engines_u, self.engines_c, self.engines_m
            main = torch.cuda.current_stream()
            side = self._sides[0] if self.two_streams else None
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ec.rec_head.run()
                eu.rec_head.run()
                main.wait_stream(side)
            else:
                eu.rec_head.run()
                ec.rec_head.run()
            em.rec.run()
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ec.rec_tail.run()
                eu.rec_tail.run()
                main.wait_stream(side)
            else:
                eu.rec_tail.run()
                ec.rec_tail.run()
            self.tail.run()
            return
        if self.two_streams:
            # the unconditional and conditional forwards are independent until the CFG combine: fork them onto two HIP
            # streams (two parallel branches of the captured graph) so the small low-resolution launches of one overlap
            # the other's; joined before the combine
            main = torch.cuda.current_stream()
            engines = self.engines_u + self.engines_c
            for side, eng in zip(self._sides, engines[1:]):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    eng.rec.run()
            engines[0].rec.run()
            for side in self._sides:
                main.wait_stream(side)
        else:
            for e in self.engines_u + self.engines_c:
                e.rec.run()
        self.tail.run()

    def capture(self):
        """Capture one step into a HIP graph.  The capture itself does not execute the step."""
        if self.graph is not None:
            return
        torch.cuda.synchronize()
        s = torch.cuda.Stream(device=self.latents.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):      # warm the kernels (lazy module load, hipFuncSetAttribute) outside capture
            keep = (self.latents.clone(), self.x0_prev.clone(), self.state.clone())
            self._step_eager()
            self.latents.copy_(keep[0]); self.x0_prev.copy_(keep[1]); self.state.copy_(keep[2])
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_eager()
        self.graph = g

    def step(self):
        # the device tables have T rows: a step past the end of the schedule is a caller bug (the kernels clamp the index, so it
        # could not read out of bounds, but the result would be meaningless)
        if self._host_step >= self.T:
            raise RuntimeError(f"DenoiseLoop.step(): all {self.T} steps of the schedule have run; call reset() first")
        self._host_step += 1
        if self.use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self._step_eager()

    def run(self, steps: Optional[int] = None) -> torch.Tensor:
        for _ in range(self.T if steps is None else steps):
            self.step()
        return self.latents


def shard_batch(total: int, rank: int, world: int) -> slice:
    """Contiguous, even split of the global batch (ranks must divide it)."""
    if total % world:
        raise ValueError(f"global batch {total} is not divisible by world size {world}")
    per = total // world
    return slice(rank * per, (rank + 1) * per)


def gather_latents(local: torch.Tensor, world: int, force: bool = False) -> torch.Tensor:
    """The single collective of the multi-GPU path: all_gather of the final latents (RCCL over xGMI on GPUs,
    gloo in the CPU tests).  Rank order == batch order, so the result equals the 1-GPU run sample for sample."""
    import torch.distributed as dist
    if world == 1 and not force:
        return local
    src = local.contiguous()
    if src.is_cuda and dist.get_backend() == "gloo":     # gloo has no device all_gather (tests on a 1-GPU box): stage through the host
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out.to(local.device)


class PhotoVersePipeline:
    """Convenience bundle named in the task text (the reference itself has no such class): holds what
    ``load_models`` returns and calls ``run_inference`` - optionally sharding the batch over the ranks of an initialised
    ``torch.distributed`` process group and gathering the final latents with one collective."""

    def __init__(self, tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, lora_config=None):
        self.tokenizer, self.text_encoder, self.vae, self.unet = tokenizer, text_encoder, vae, unet
        self.image_encoder, self.image_adapter, self.text_adapter = image_encoder, image_adapter, text_adapter
        self.scheduler, self.lora_config = scheduler, lora_config
        self.device = torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, extra_num_tokens=4, photoverse_path=None, **kw):
        from .modeling_utils import load_models
        return cls(*load_models(pretrained_model_name_or_path, extra_num_tokens, photoverse_path, **kw))

    def to(self, device):
        self.device = torch.device(device)
        for m in (self.unet, self.vae, self.text_encoder, self.image_encoder, self.image_adapter, self.text_adapter):
            if m is not None:
                m.to(self.device)
        return self

    @torch.no_grad()
    def __call__(self, example, image_encoder_layers_idx=(4, 8, 12, 16), shard: bool = False, **kw):
        from .infer import run_inference
        if shard:
            import torch.distributed as dist
            rank, world = dist.get_rank(), dist.get_world_size()
            n = example["pixel_values_clip"].shape[0]
            sl = shard_batch(n, rank, world)
            local = {k: (v[sl] if torch.is_tensor(v) and v.shape[:1] == (n,) else v) for k, v in example.items()}
            if kw.get("seed") is not None:
                # the noise of the GLOBAL batch is drawn once with the reference's generator semantics (infer.py:52-59: CPU global
                # generator) on every rank and sliced, so sample i equals sample i of the seeded 1-GPU run
                latent_size = kw.get("latent_size", 64)
                generator = torch.manual_seed(kw["seed"])
                kw["noise"] = torch.randn((n, self.unet.config.in_channels, latent_size, latent_size), generator=generator)[sl]
            out = run_inference(local, self.tokenizer, self.image_encoder, self.text_encoder, self.unet, self.text_adapter,
                                self.image_adapter, self.vae, self.scheduler, self.device, list(image_encoder_layers_idx), **kw)
            return gather_latents(out, world, force=True)
        return run_inference(example, self.tokenizer, self.image_encoder, self.text_encoder, self.unet, self.text_adapter,
                             self.image_adapter, self.vae, self.scheduler, self.device, list(image_encoder_layers_idx), **kw)
