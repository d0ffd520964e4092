"""One PhotoVerse training step (``/root/reference/train.py:466-545``) on the HIP kernels.

``training_step_forward``: what an iteration evaluates before ``accelerator.backward(loss)`` - VAE-encode the pixels and sample the
posterior (``:473-474``), draw noise and per-sample timesteps and add the noise (``:477-484``), CLIP image features -> both adapters in
FULL mode (all ``extra_num_tokens + 1`` tokens, ``:487-502``), the dict-input text encoder with concept injection (``:497-499``), the UNet
in grad mode - per-sample timesteps ``(B,)``, every cross-attention layer drawing its branch fusion
(``attention_processor.py:413-420``; here on the device, ``pv_fusion_draw``) - and the three loss terms (``:509-516,541``):

    loss = mse(noise_pred, noise) + 0.01 * mean|concept_text_embeddings| + 0.001 * mean(to_v_ip_norm stack)   [+ 0.01 * face loss]

on the inference-style engine (fused GEGLU / fused attn2, nothing kept for a backward).

``TrainStep`` (below): the trainable part of the step - adapters, text encoder, UNet, losses - as a forward AND a backward launch plan
(``tape.py``), leaving gradients on every parameter ``train.py:366-377`` optimises; ``optim.AdamW`` finishes the iteration
(``:538-545``).  ``lora_dropout > 0`` (the reference default, 0.1) runs the low-rank branches un-merged with a device-side dropout.  ``face_loss=`` adds
the ArcFace identity-loss branch (``:521-535``); ``training_iteration`` is the loop body of ``:464-549``.
"""
from __future__ import annotations

from typing import Optional

import torch

from .ops import Recorder


def _train_engine(unet, batch, h, w, n_ip, fusion_seed):
    cache = unet.__dict__.setdefault("_train_engines", {})
    key = (batch, h, w, n_ip, int(fusion_seed), unet.__dict__.get("_pack_version", 0))
    eng = cache.get(key)
    if eng is None:
        cache.clear()                                    # one training shape at a time: an engine holds ~2 GB of activations at bs=16
        eng = cache[key] = unet.engine(batch, h, w, n_ip, batch, device_fusion="always", fusion_seed=fusion_seed)
    return eng


@torch.no_grad()
def training_step_forward(batch, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, noise_scheduler, device,
                          image_encoder_layers_idx, extra_num_tokens: int, *, generator: Optional[torch.Generator] = None, fusion_seed: int = 0,
                          forced_fusion=None, noise=None, timesteps=None, posterior_eps=None):
    """Returns a dict: ``loss``, ``diffusion_loss``, ``concept_text_loss``, ``cross_attn_visual_loss`` (fp32 device scalars, shape
    (1,)), ``noise_pred`` and the draws it used (``noise``, ``timesteps``, ``latents``).  ``noise`` / ``timesteps`` / ``posterior_eps``
    / ``forced_fusion`` (per cross-attention layer u in [0, 1), in ``engine.fusion_names`` order) override the random draws (tests)."""
    device = torch.device(device)
    pixel_values = batch["pixel_values"].to(device, dtype=torch.float32)
    pixel_values_clip = batch["pixel_values_clip"].to(device, dtype=torch.float32)
    placeholder_idx = batch["concept_placeholder_idx"].to(device)
    text_input_ids = batch["text_input_ids"].to(device)
    bsz = pixel_values.shape[0]

    # train.py:473-474 - latents = vae.encode(x).latent_dist.sample() * scaling_factor
    dist = vae.encode(pixel_values).latent_dist
    rec = Recorder(device)
    if posterior_eps is None:
        posterior_eps = torch.randn(dist.mean.shape, generator=generator, device="cpu" if generator is not None else device)
    sample = rec.posterior_sample(dist.parameters.contiguous(), posterior_eps.to(device, torch.float32).contiguous())
    sf = torch.full((bsz,), float(vae.config.scaling_factor), dtype=torch.float32, device=device)
    latents = rec.affine_rows(sample, sf)
    rec.run()

    # train.py:477-484 - noise, per-sample timesteps, forward diffusion
    if noise is None:
        noise = torch.randn(latents.shape, generator=generator, device="cpu" if generator is not None else device)
    noise = noise.to(device, torch.float32).contiguous()
    n_train = noise_scheduler.config["num_train_timesteps"] if isinstance(noise_scheduler.config, dict) else noise_scheduler.config.num_train_timesteps
    if timesteps is None:
        timesteps = torch.randint(0, n_train, (bsz,), generator=generator)
    timesteps = timesteps.long().cpu()
    from .scheduler import DPMSolverMultistepScheduler
    sch = noise_scheduler if hasattr(noise_scheduler, "add_noise") else DPMSolverMultistepScheduler.from_config(noise_scheduler.config)
    noisy_latents = sch.add_noise(latents, noise, timesteps)

    # train.py:487-502 - image features -> adapters (all tokens), text encoder with the injected concept embeddings
    image_features = image_encoder(pixel_values_clip, output_hidden_states=True)
    image_embeddings = [image_features[0]] + [image_features[2][i] for i in image_encoder_layers_idx if i < len(image_features[2])]
    assert len(image_embeddings) == extra_num_tokens + 1, "Entered indices are out of range for image_encoder layers."
    concept_text_embeddings = text_adapter(image_embeddings)
    encoder_hidden_states = text_encoder({"text_input_ids": text_input_ids, "concept_text_embeddings": concept_text_embeddings,
                                          "concept_placeholder_idx": placeholder_idx})[0]
    encoder_hidden_states_image = image_adapter(image_embeddings)

    # train.py:505-506 - the UNet in grad mode: per-sample timesteps, per-layer branch fusion drawn on the device
    B, _, h, w = noisy_latents.shape
    P = encoder_hidden_states_image.shape[1]
    eng = _train_engine(unet, B, h, w, P, fusion_seed)
    eng.x_in.copy_(noisy_latents)
    eng.text.copy_(encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[-1]))
    eng.ip.copy_(encoder_hidden_states_image.reshape(-1, encoder_hidden_states_image.shape[-1]))
    eng.timesteps.copy_(timesteps.to(device, torch.float32))
    if forced_fusion is not None:
        eng.fusion_forced.copy_(torch.as_tensor(forced_fusion, dtype=torch.float32))
    else:
        eng.fusion_forced.fill_(-1.0)
    noise_pred = eng.run()

    # train.py:509-516, 541 - regularisers and the loss
    rec = Recorder(device)
    concept_text_loss = rec.reduce_mean(concept_text_embeddings.contiguous(), mode="abs")
    per_layer = rec.empty((len(eng.vnorms),), torch.float32)
    for i, name in enumerate(eng.vnorms):                       # get_visual_cross_attention_values_norm(unet).mean(): equal-sized layers
        rec.reduce_mean(eng.vnorms[name], mode="mean", out=per_layer[i:i + 1])
    cross_attn_visual_loss = rec.reduce_mean(per_layer, mode="mean")
    diffusion_loss = rec.reduce_mean(noise_pred.contiguous(), noise, mode="mse")
    terms = rec.hold(torch.zeros(3, dtype=torch.float32, device=device))
    rec.run()
    # loss = diffusion + 0.01 * concept + 0.001 * visual: three scalars, combined by the same row-affine kernel
    terms[0:1].copy_(diffusion_loss); terms[1:2].copy_(concept_text_loss); terms[2:3].copy_(cross_attn_visual_loss)
    rec2 = Recorder(device)
    w = torch.tensor([3.0, 0.03, 0.003], dtype=torch.float32, device=device)     # mean of the three weighted terms == 1 * d + 0.01 * c + 0.001 * v
    weighted = rec2.affine_rows(terms.view(3, 1), w)
    loss = rec2.reduce_mean(weighted.view(-1), mode="mean")
    rec2.run()
    return {"loss": loss, "diffusion_loss": diffusion_loss, "concept_text_loss": concept_text_loss,
            "cross_attn_visual_loss": cross_attn_visual_loss, "noise_pred": noise_pred, "noise": noise, "timesteps": timesteps,
            "latents": latents, "fusion_table": eng.fusion_tab.clone(), "fusion_names": list(eng.fusion_names)}


# ======================================================================================================================
# The full step: forward + backward as two static launch plans (tape.py), AdamW on the device (optim.py)
# ======================================================================================================================
from types import SimpleNamespace  # noqa: E402
from typing import Dict, List  # noqa: E402

from . import ops as _ops  # noqa: E402
from .lora import LoRALinear  # noqa: E402
from .tape import Tape, Var, conv3_dgrad_weight  # noqa: E402
from .unet import ResnetBlock2D, Transformer2DModel, _conv1_w, _conv3_w, _f16, _f32  # noqa: E402


class TrainStep:
    """One PhotoVerse training iteration (``/root/reference/train.py:466-545``; the ArcFace identity-loss term of ``:521-535`` with
    ``face_loss=``) for a fixed shape, as forward and backward launch plans over the HIP kernels.

    Trainable (train.py:366-377): both adapters, ``to_k_ip`` / ``to_v_ip`` of the 16 cross-attention processors and - with
    ``--use_lora`` - the LoRA factors behind ``attn2.to_q / to_k / to_v``.  Everything else is frozen, but the gradient crosses it:
    the whole UNet (dX of every conv / Linear / GroupNorm / LayerNorm / attention / GEGLU, skip connections, down / up sampling) and
    the 12 CLIP text layers between the injected concept rows and the text K / V projections.

    ``step(batch)`` fills the static inputs, replays the two plans and leaves ``.grad`` (times ``grad_scale``, fp32) on every
    trainable parameter; ``optim.AdamW.step(clip_groups=..., grad_scale=...)`` finishes the iteration without a host sync.
    With ``lora_dropout == 0`` LoRA runs merged and the factor gradients dA = s B^T dW, dB = s dW A^T are rank-r products of the
    merged-weight gradient (library GEMM); with dropout the low-rank branches run un-merged on the HIP GEMM (``_lora_branch``).
    """

    def __init__(self, unet, text_encoder, text_adapter, image_adapter, *, batch: int, h: int, w: int, n_tokens: int, seq: int = 77,
                 clip_tokens: int = 257, clip_dim: int = 1024, grad_scale: float = 4096.0, fusion_seed: int = 0,
                 loss_weights=(1.0, 0.01, 0.001), use_graph: bool = True, face_loss=None, vae=None, noise_scheduler=None,
                 face_samples: Optional[int] = None, face_weight: float = 0.01, guidance_scale: float = 2.0, infer_steps: int = 10,
                 image_size: Optional[int] = None):
        """``face_loss`` (a ``loss.FaceLoss``) + ``vae`` switch on the identity-loss branch of train.py:521-535: for ``face_samples``
        images of the batch (default ``max(int(0.25 * batch), 1)``, train.py:247,522) ``run_inference(from_noised_image=True,
        training_mode=True, timesteps=infer_steps, guidance_scale=..., token_index=0)`` is replayed - the first ``infer_steps - 1``
        denoising steps without gradient, the last one, the VAE decode and the ArcFace trunk inside the differentiated plan."""
        dev = unet.device
        self.use_graph, self.graph, self._warm = bool(use_graph), None, False
        self.face_loss, self.vae, self.face = face_loss, vae, None
        self.unet, self.text_encoder, self.text_adapter, self.image_adapter = unet, text_encoder, text_adapter, image_adapter
        self.B, self.H, self.W, self.E, self.S_len = batch, h, w, n_tokens, seq
        self.grad_scale = float(grad_scale)
        self.loss_weights = tuple(float(x) for x in loss_weights)
        tp = self.tape = Tape(dev, grad_scale)
        rf = tp.rf
        cfg = unet.config
        self.pgrads: List[tuple] = []
        # static inputs
        self.noisy = rf.hold(torch.zeros((batch, cfg.in_channels, h, w), dtype=torch.float32, device=dev))
        self.noise = rf.hold(torch.zeros_like(self.noisy))
        self.timesteps = rf.hold(torch.zeros((batch,), dtype=torch.float32, device=dev))
        self.ids = rf.hold(torch.zeros((batch, seq), dtype=torch.int64, device=dev))
        self.pidx = rf.hold(torch.zeros((batch,), dtype=torch.int64, device=dev))
        self.pidx32 = rf.hold(torch.zeros((batch,), dtype=torch.int32, device=dev))
        self.embs = [rf.hold(torch.zeros((batch, clip_tokens, clip_dim), dtype=torch.float16, device=dev)) for _ in range(n_tokens)]
        self.t_state = None                               # per-sample timesteps (a loop pass reads one timestep through a step counter)
        self._build(fusion_seed)
        if face_loss is not None:
            if vae is None:
                raise ValueError("the face-loss branch needs the VAE (run_inference decodes the generated latents, infer.py:121-123)")
            self._build_face(face_samples if face_samples is not None else max(int(0.25 * batch), 1), float(face_weight), float(guidance_scale),
                             int(infer_steps), noise_scheduler, fusion_seed, clip_tokens, clip_dim, image_size or 8 * h)

    # ------------------------------------------------------------------ parameter-gradient sinks
    def _to(self, param):
        def sink(buf):
            self.pgrads.append((param, buf))
        return sink

    def _affine_to(self, ln):
        def sink(dgb):
            self.pgrads.append((ln.weight, dgb[0]))
            self.pgrads.append((ln.bias, dgb[1]))
        return sink

    # ------------------------------------------------------------------ adapters (adapters.py:30-44, full mode)
    def _mlp(self, seq, x: Var, group: int) -> Var:
        tp = self.tape

        def lin(i, xin):
            L = seq[i]
            w16, wT = tp.trainable_blocks(L.out_features, L.in_features, [(L.weight, 0, 0, 1.0)])
            return tp.linear(xin, w16, wT, bias=L.bias.data, on_wgrad=self._to(L.weight), on_bgrad=self._to(L.bias))
        h0 = lin(0, x)
        a1 = tp.layernorm(h0, seq[1].weight.data, seq[1].bias.data, eps=seq[1].eps, act=_ops.ACT_LEAKY_RELU, on_affine=self._affine_to(seq[1]))
        h3 = lin(3, a1)
        a4 = tp.layernorm(h3, seq[4].weight.data, seq[4].bias.data, eps=seq[4].eps, act=_ops.ACT_LEAKY_RELU, on_affine=self._affine_to(seq[4]),
                          mean_group=group)
        return lin(6, a4)

    def _adapter(self, adapter, embs=None) -> Var:
        """Tokens 0 .. len(embs)-1 of the adapter (adapters.py:30-44: all of them in the training forward, token_index=0 alone inside
        run_inference)."""
        tp, B = self.tape, self.B
        embs = self.embs if embs is None else embs
        cout = getattr(adapter, "mapping_0")[6].out_features
        whole = tp.rf.empty((B, len(embs) * cout))
        parts = []
        for i, emb in enumerate(embs):
            T, D = emb.shape[1], emb.shape[2]
            cls = self._mlp(getattr(adapter, f"mapping_{i}"), Var(emb.view(B, T * D)[:, :D]), 1)
            pat = self._mlp(getattr(adapter, f"mapping_patch_{i}"), Var(emb.view(B * T, D)), T)
            parts.append(tp.add_into(cls, pat, whole[:, i * cout:(i + 1) * cout]))
        return tp.columns(parts, whole)

    # ------------------------------------------------------------------ text encoder (clip.py:29-102 with the injected concept rows)
    def _text(self, concept: Var) -> Var:
        tp, B, S, E = self.tape, self.B, self.S_len, self.E
        te = self.text_encoder
        cfg, tm = te.config, te.text_model
        dim, heads = cfg.hidden_size, cfg.num_attention_heads
        x0 = tp.rf.clip_text_embed(self.ids, _f32(tm.embeddings.token_embedding.weight), _f32(tm.embeddings.position_embedding.weight),
                                   concept.t.view(B * E, dim), self.pidx, n_concept=E, batch=B, seq=S, dim=dim)
        x = Var(x0, True)
        pidx32 = self.pidx32

        def embed_bwd(x=x):
            if x.g is None:
                return
            g32 = tp.rb.gather_rows(x.g, pidx32, batch=B, seq=S, n_e=E)
            tp._accum(concept, tp.rb.cast_to_f16(g32.view(B, E * dim)))
        tp.back.append(embed_bwd)
        for lyr in tm.encoder.layers:
            sa = lyr.self_attn
            hcur = tp.layernorm(x, _f32(lyr.layer_norm1.weight), _f32(lyr.layer_norm1.bias), eps=lyr.layer_norm1.eps)
            wqkv = torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0)
            bqkv = torch.cat([_f32(sa.q_proj.bias), _f32(sa.k_proj.bias), _f32(sa.v_proj.bias)], 0).contiguous()
            qkv = tp.linear(hcur, *tp.frozen(wqkv), bias=bqkv, rows_per_image=S)
            a = tp.self_attention(qkv, batch=B, heads=heads, n=S, d=dim // heads, causal=True)
            x = tp.linear(a, *tp.frozen(sa.out_proj.weight), bias=_f32(sa.out_proj.bias), residual=x, rows_per_image=S)
            hcur = tp.layernorm(x, _f32(lyr.layer_norm2.weight), _f32(lyr.layer_norm2.bias), eps=lyr.layer_norm2.eps)
            f = tp.linear(hcur, *tp.frozen(lyr.mlp.fc1.weight), bias=_f32(lyr.mlp.fc1.bias), rows_per_image=S)
            f = tp.activation(f, _ops.ACT_QUICK_GELU)
            x = tp.linear(f, *tp.frozen(lyr.mlp.fc2.weight), bias=_f32(lyr.mlp.fc2.bias), residual=x, rows_per_image=S)
        return tp.layernorm(x, _f32(tm.final_layer_norm.weight), _f32(tm.final_layer_norm.bias), eps=tm.final_layer_norm.eps)

    # ------------------------------------------------------------------ UNet blocks ([EXT] diffusers; launch plan as unet.UNetEngine)
    def _resnet(self, m: ResnetBlock2D, x: Var, x1: Optional[Var], h, w, temb_all, toff) -> Var:
        tp, B = self.tape, self.B
        cout = m.conv1.out_channels
        hn = tp.groupnorm(x, _f32(m.norm1.weight), _f32(m.norm1.bias), batch=B, hw=h * w, x1=x1, eps=m.norm1.eps, act=_ops.ACT_SILU)
        h1 = tp.conv3(hn, _conv3_w(m.conv1.weight), conv3_dgrad_weight(m.conv1.weight), bias=_f32(m.conv1.bias), batch=B, h=h, w=w,
                      rowadd=temb_all[:, toff:toff + cout], rowadd_ld=temb_all.stride(0) if temb_all.shape[0] > 1 else 0)
        h2 = tp.groupnorm(h1, _f32(m.norm2.weight), _f32(m.norm2.bias), batch=B, hw=h * w, eps=m.norm2.eps, act=_ops.ACT_SILU)
        if m.conv_shortcut is not None:
            sc = tp.linear(x, *tp.frozen(_conv1_w(m.conv_shortcut.weight)), bias=_f32(m.conv_shortcut.bias), x1=x1, rows_per_image=h * w)
        else:
            assert x1 is None
            sc = x
        return tp.conv3(h2, _conv3_w(m.conv2.weight), conv3_dgrad_weight(m.conv2.weight), bias=_f32(m.conv2.bias), batch=B, h=h, w=w, residual=sc)

    def _maybe_lora(self, lin_mod):
        """(w16, wT16, sink) of an attn2 projection: trainable through its LoRA factors, else frozen."""
        tp = self.tape
        if isinstance(lin_mod, LoRALinear):
            return tp.trainable(lambda m=lin_mod: m.weight), lin_mod
        return tp.frozen(lin_mod.weight), None

    def _lora_branch(self, x: Var, mods, site: int, rows_per_image) -> Var:
        """peft's un-merged LoRA forward for ``len(mods)`` projections of the same input (train.py:348-354 with lora_dropout > 0):
        y = [W_i x]_i + [s_i B_i A_i dropout_i(x)]_i.  The low-rank factors are zero-padded to 128 rows / columns per projection (the MFMA
        GEMM's tile width) and stacked block-diagonally, so the branch is: one dropout launch (``len(mods)`` independently masked copies
        side by side), one GEMM to the stacked rank space, one GEMM back with the base projection as its residual."""
        tp = self.tape
        n, RP = len(mods), 128
        cin = mods[0].in_features
        couts = [m.out_features for m in mods]
        p = mods[0].dropout_p
        for m in mods:
            if m.lora_A["default"].weight.shape[0] > RP:
                raise NotImplementedError("LoRA rank > 128")
        base = tp.linear(x, *tp.frozen(torch.cat([m.base_layer.weight for m in mods], 0)), rows_per_image=rows_per_image)
        xd = tp.dropout(x, p=p, rng=self.fusion_rng, site=site, copies=n)                       # [M, n * cin]
        self.dropout_sites.append((site, n, cin, p))

        # block-diagonal stacks, re-made from the fp32 factors every step: A_i at (i * 128, i * cin), s_i * B_i at (row offset of i, i * 128)
        a_bd = tp.trainable_blocks(n * RP, n * cin, [(m.lora_A["default"].weight, i * RP, i * cin, 1.0) for i, m in enumerate(mods)])
        b_bd = tp.trainable_blocks(sum(couts), n * RP, [(m.lora_B["default"].weight, sum(couts[:i]), i * RP, m.scaling) for i, m in enumerate(mods)])

        def sink_a(dW):                                   # [n * 128, n * cin]
            for i, m in enumerate(mods):
                A = m.lora_A["default"].weight
                self.pgrads.append((A, lambda dW=dW, i=i, A=A: dW[i * RP:i * RP + A.shape[0], i * cin:(i + 1) * cin].contiguous()))

        def sink_b(dW):                                   # [sum couts, n * 128]; B_pad = s * B -> dB = s * dB_pad (s applied by the slab sum: wgrad_scale)
            r0 = 0
            for i, m in enumerate(mods):
                Bm = m.lora_B["default"].weight
                self.pgrads.append((Bm, lambda dW=dW, i=i, r0=r0, Bm=Bm: dW[r0:r0 + couts[i], i * RP:i * RP + Bm.shape[1]].contiguous()))
                r0 += couts[i]
        u = tp.linear(xd, *a_bd, rows_per_image=rows_per_image, on_wgrad=sink_a)                 # [M, n * 128]
        if len({float(m.scaling) for m in mods}) != 1:
            raise NotImplementedError("LoRA branches of one site with different scalings")
        return tp.linear(u, *b_bd, residual=base, rows_per_image=rows_per_image, on_wgrad=sink_b, wgrad_scale=float(mods[0].scaling))

    def _lora_sink(self, mods_rows):
        """mods_rows: [(LoRALinear or None, row0, row1)] - slices of a stacked merged-weight gradient."""
        def sink(dW):
            for mod, r0, r1 in mods_rows:
                if mod is not None:
                    self.lora_pending.append((mod, dW[r0:r1]))
        return sink

    def _transformer(self, name: str, m: Transformer2DModel, x: Var, h, w, text: Var, ip: Var) -> Var:
        tp, B = self.tape, self.B
        n = h * w
        blk = m.transformer_blocks[0]
        C = m.proj_in.out_channels
        heads = blk.attn1.heads
        d = C // heads
        g = tp.groupnorm(x, _f32(m.norm.weight), _f32(m.norm.bias), batch=B, hw=n, eps=m.norm.eps, act=_ops.ACT_NONE)
        hs = tp.linear(g, *tp.frozen(_conv1_w(m.proj_in.weight)), bias=_f32(m.proj_in.bias), rows_per_image=n)
        # attn1 ([EXT] AttnProcessor2_0, models/unet.py:20-24)
        a1 = blk.attn1
        qkv = tp.ln_linear(hs, _f32(blk.norm1.weight), _f32(blk.norm1.bias), *tp.frozen(torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0)),
                           eps=blk.norm1.eps, rows_per_image=n)
        sa = tp.self_attention(qkv, batch=B, heads=heads, n=n, d=d)
        hs = tp.linear(sa, *tp.frozen(a1.to_out[0].weight), bias=_f32(a1.to_out[0].bias), residual=hs, rows_per_image=n)
        # attn2 (PhotoVerseAttnProcessor2_0, attention_processor.py:245-435), grad mode: fusion drawn on the device
        a2, proc = blk.attn2, blk.attn2.processor
        n2 = tp.layernorm(hs, _f32(blk.norm2.weight), _f32(blk.norm2.bias), eps=blk.norm2.eps)
        lk = a2.to_k if isinstance(a2.to_k, LoRALinear) else None
        lv = a2.to_v if isinstance(a2.to_v, LoRALinear) else None
        site = self.site_base + 4 * len(self.fusion_names)   # dropout stream ids of this layer: site (q), site + 1 (k and v: copies 0 / 1)
        if isinstance(a2.to_q, LoRALinear) and a2.to_q.dropout_p > 0:
            q = self._lora_branch(n2, [a2.to_q], site, rows_per_image=n)
        else:
            (wq, wqT), lq = self._maybe_lora(a2.to_q)
            q = tp.linear(n2, wq, wqT, rows_per_image=n, on_wgrad=self._lora_sink([(lq, 0, C)]) if lq is not None else None)
        if lk is not None and lv is not None and (lk.dropout_p > 0 or lv.dropout_p > 0):
            if lk.dropout_p != lv.dropout_p:
                raise NotImplementedError("to_k / to_v LoRA layers with different dropout rates")
            kvt = self._lora_branch(text, [lk, lv], site + 1, rows_per_image=self.S_len)
        else:
            if any(m is not None and m.dropout_p > 0 for m in (lk, lv)):
                raise NotImplementedError("lora_dropout on only one of attn2.to_k / attn2.to_v")
            kv_fn = lambda a2=a2: torch.cat([a2.to_k.weight, a2.to_v.weight], 0)
            wkv, wkvT = tp.trainable(kv_fn) if (lk is not None or lv is not None) else tp.frozen(kv_fn())
            kvt = tp.linear(text, wkv, wkvT, rows_per_image=self.S_len,
                            on_wgrad=self._lora_sink([(lk, 0, C), (lv, C, 2 * C)]) if (lk is not None or lv is not None) else None)
        kip, vip = proc.to_k_ip[0], proc.to_v_ip[0]
        wkvip, wkvipT = tp.trainable_blocks(kip.out_features + vip.out_features, kip.in_features,
                                            [(kip.weight, 0, 0, 1.0), (vip.weight, kip.out_features, 0, 1.0)])

        def ip_sink(dW, kip=kip, vip=vip, C=C):
            self.pgrads.append((kip.weight, dW[:C]))
            self.pgrads.append((vip.weight, dW[C:]))
        kvip = tp.linear(ip, wkvip, wkvipT, rows_per_image=self.E, on_wgrad=ip_sink)
        vnorm = tp.rf.empty((B, heads, self.E), torch.float32)
        self.vnorms[name] = vnorm
        fus = self.fusion_tab[len(self.fusion_names)] if self.fusion_tab is not None else None     # None: (1, 1), the no_grad rule (:411-412)
        self.fusion_names.append(name)
        xa = tp.cross_attention(q, kvt, kvip, batch=B, heads=heads, n=n, nt=self.S_len, nip=self.E, d=d, fusion=fus, vnorm=vnorm,
                                vnorm_coef=self.vnorm_weight * self.grad_scale / (self.n_xattn * B * heads * self.E))
        hs = tp.linear(xa, *tp.frozen(a2.to_out[0].weight), bias=_f32(a2.to_out[0].bias), residual=hs, rows_per_image=n)
        # GEGLU feed-forward (pre-activation kept for the backward)
        hg = tp.ln_linear(hs, _f32(blk.norm3.weight), _f32(blk.norm3.bias), *tp.frozen(blk.ff.net[0].proj.weight), eps=blk.norm3.eps,
                          bias=_f32(blk.ff.net[0].proj.bias), rows_per_image=n)
        gg = tp.geglu(hg)
        hs = tp.linear(gg, *tp.frozen(blk.ff.net[2].weight), bias=_f32(blk.ff.net[2].bias), residual=hs, rows_per_image=n)
        return tp.linear(hs, *tp.frozen(_conv1_w(m.proj_out.weight)), bias=_f32(m.proj_out.bias), residual=x, rows_per_image=n, colstats=True)

    # ------------------------------------------------------------------ the plan
    def _unet_pass(self, noisy: torch.Tensor, text: Var, ip_rows: Var):
        """One UNet forward on the current tape (``noisy``: fp32 (B, 4, h, w) buffer; timesteps from ``self.timesteps`` or, in a loop pass,
        from ``self.t_state``).  Returns (noise_pred fp32 NCHW, seed): ``seed(dpred)`` - called while the backward plan is built -
        turns the fp32 gradient of the prediction into the gradient of conv_out's input.  No gradient is computed upstream of the first
        cross-attention layer."""
        tp, u, B = self.tape, self.unet, self.B
        rf, rb = tp.rf, tp.rb
        dev = tp.device
        cfg = u.config
        h, w = self.H, self.W
        c0 = cfg.block_out_channels[0]
        if self.t_state is not None:                      # a loop pass: one timestep for all rows, read through the device step counter
            te = rf.timestep_embedding(self.t_state[0], self.t_state[1], 1, c0)
        else:
            te = rf.timestep_embedding(self.timesteps, None, B, c0)
        e1 = rf.gemm(te, _f16(u.time_embedding.linear_1.weight), bias=_f32(u.time_embedding.linear_1.bias), act=_ops.ACT_SILU)
        e2 = rf.gemm(e1, _f16(u.time_embedding.linear_2.weight), bias=_f32(u.time_embedding.linear_2.bias), act=_ops.ACT_SILU)
        resnets = [m for m in u.modules() if isinstance(m, ResnetBlock2D)]
        toffs, off = {}, 0
        for m in resnets:
            toffs[id(m)] = off
            off += m.conv1.out_channels
        pad = (-off) % 160
        wt = torch.cat([_f16(m.time_emb_proj.weight) for m in resnets] +
                       ([torch.zeros(pad, resnets[0].time_emb_proj.in_features, dtype=torch.float16, device=dev)] if pad else []), 0)
        bt = torch.cat([_f32(m.time_emb_proj.bias) for m in resnets] + ([torch.zeros(pad, device=dev)] if pad else []), 0)
        temb_all = rf.gemm(e2, wt.contiguous(), bias=bt.contiguous(), out_f32=True)
        kin = cfg.in_channels * 9
        kpad = (kin + 63) // 64 * 64
        cols = rf.im2col3x3(noisy, batch=B, cin=cfg.in_channels, h=h, wd=w, kpad=kpad)
        w_in = torch.zeros(c0, kpad, dtype=torch.float16, device=dev)
        w_in[:, :kin] = u.conv_in.weight.detach().reshape(c0, kin).to(torch.float16)
        x = Var(rf.gemm(cols, w_in, bias=_f32(u.conv_in.bias), rows_per_image=h * w, colstats=True), False)
        skips = [(x, h, w)]
        for bi, blk in enumerate(u.down_blocks):
            for i, res in enumerate(blk.resnets):
                x = self._resnet(res, x, None, h, w, temb_all, toffs[id(res)])
                if blk.has_attn:
                    x = self._transformer(f"down_blocks.{bi}.attentions.{i}", blk.attentions[i], x, h, w, text, ip_rows)
                skips.append((x, h, w))
            if blk.downsamplers is not None:
                conv = blk.downsamplers[0].conv
                x = tp.conv3(x, _conv3_w(conv.weight), conv3_dgrad_weight(conv.weight), bias=_f32(conv.bias), batch=B, h=h, w=w, stride=2)
                h, w = h // 2, w // 2
                skips.append((x, h, w))
        mb = u.mid_block
        x = self._resnet(mb.resnets[0], x, None, h, w, temb_all, toffs[id(mb.resnets[0])])
        x = self._transformer("mid_block.attentions.0", mb.attentions[0], x, h, w, text, ip_rows)
        x = self._resnet(mb.resnets[1], x, None, h, w, temb_all, toffs[id(mb.resnets[1])])
        for bi, blk in enumerate(u.up_blocks):
            for i, res in enumerate(blk.resnets):
                sk, sh, sw = skips.pop()
                assert (sh, sw) == (h, w)
                x = self._resnet(res, x, sk, h, w, temb_all, toffs[id(res)])
                if blk.has_attn:
                    x = self._transformer(f"up_blocks.{bi}.attentions.{i}", blk.attentions[i], x, h, w, text, ip_rows)
            if blk.upsamplers is not None:
                conv = blk.upsamplers[0].conv
                x = tp.conv3(x, _conv3_w(conv.weight), conv3_dgrad_weight(conv.weight), bias=_f32(conv.bias), batch=B, h=h, w=w, upsample=1)
                h, w = h * 2, w * 2
        xn = tp.groupnorm(x, _f32(u.conv_norm_out.weight), _f32(u.conv_norm_out.bias), batch=B, hw=h * w, eps=u.conv_norm_out.eps, act=_ops.ACT_SILU)
        wo = u.conv_out.weight.detach().permute(0, 2, 3, 1).reshape(cfg.out_channels, -1).to(torch.float16).contiguous()
        noise_pred = rf.conv_out(xn.t, wo, _f32(u.conv_out.bias), batch=B, cin=c0, h=h, wd=w, cout=cfg.out_channels)


        def seed(dpred):
            oc = cfg.out_channels
            kp = (oc * 9 + 63) // 64 * 64
            dcols = rb.im2col3x3(dpred, batch=B, cin=oc, h=h, wd=w, kpad=kp)
            wd_out = torch.zeros(c0, kp, dtype=torch.float16, device=dev)
            wd_out[:, :oc * 9] = u.conv_out.weight.detach().flip(2, 3).permute(1, 0, 2, 3).reshape(c0, oc * 9).to(torch.float16)
            xn.g = rb.gemm(dcols, wd_out, rows_per_image=h * w)
        return noise_pred, seed

    def _build(self, fusion_seed: int):
        import numpy as np
        from .attention_processor import PhotoVerseAttnProcessor2_0
        from .unet import Attention
        tp, u, B = self.tape, self.unet, self.B
        rf, rb = tp.rf, tp.rb
        dev = tp.device
        cfg = u.config
        h, w = self.H, self.W
        c0 = cfg.block_out_channels[0]
        self.vnorms: Dict[str, torch.Tensor] = {}
        self.fusion_names: List[str] = []
        self.lora_pending: List[tuple] = []
        self.dropout_sites: List[tuple] = []              # (site, copies, cols, p) of every LoRA dropout launch (tests rebuild the masks)
        self.site_base, self.vnorm_weight = 0, self.loss_weights[2]
        procs = [m.processor for _, m in u.named_modules() if isinstance(m, Attention) and isinstance(m.processor, PhotoVerseAttnProcessor2_0)]
        self.n_xattn = len(procs)
        # grad-mode branch fusion, one device-side draw per forward (attention_processor.py:413-420; pv_fusion_draw)
        self.fusion_tab = rf.hold(torch.ones((self.n_xattn, 2), dtype=torch.float32, device=dev))
        key = np.array([fusion_seed & 0xFFFFFFFF, (fusion_seed >> 32) & 0xFFFFFFFF, 0, 0], dtype=np.uint32).view(np.int32)
        self.fusion_rng = rf.hold(torch.from_numpy(key.copy()).to(dev))
        self.fusion_forced = rf.hold(torch.full((self.n_xattn,), -1.0, dtype=torch.float32, device=dev))
        p0 = procs[0]
        rf.fusion_draw(None, self.fusion_rng, self.fusion_forced, self.fusion_tab, n_layers=self.n_xattn, rule1=p0.fusion_rule1, rule2=p0.fusion_rule2,
                       scale=float(p0.scale[0]), only_last_step=False)

        # --- conditioning: adapters (trainable) and the text encoder (frozen, crossed by the gradient) - train.py:495-502
        concept = self._adapter(self.text_adapter)                      # [B, E * 768]
        self.concept = concept
        # concept-text regulariser (train.py:509): forward value + its gradient seed, in tape order right after the adapter
        self.terms = rf.hold(torch.zeros(3, dtype=torch.float32, device=dev))
        rf.reduce_mean(concept.t, mode="abs", out=self.terms[1:2])

        def concept_loss_bwd():
            c32 = rb.cast_to_f32(concept.t)
            g = rb.sign(c32, self.loss_weights[1] * self.grad_scale / concept.t.numel())
            tp._accum(concept, rb.cast_to_f16(g))
        tp.back.append(concept_loss_bwd)
        text = self._text(concept)                                      # [B * S, 768]
        ip = self._adapter(self.image_adapter)                          # [B, E * 768]  == rows [B * E, 768]
        ip_rows = Var(ip.t.view(B * self.E, -1), True)

        def ip_rows_bwd():
            if ip_rows.g is not None:                         # the sum of the 16 layers' K / V projection data gradients: a GEMM output
                assert ip_rows.g.is_contiguous()
                tp._accum(ip, ip_rows.g.view(B, -1))
        tp.back.append(ip_rows_bwd)
        self.text_states, self.ip_states = text, ip_rows

        self.noise_pred, seed_pred = self._unet_pass(self.noisy, text, ip_rows)

        # --- losses (train.py:509-516, :536): values in the forward plan, gradient seeds at the head of the backward plan
        rf.reduce_mean(self.noise_pred, self.noise, mode="mse", out=self.terms[0:1])
        per_layer = rf.empty((len(self.vnorms),), torch.float32)
        for i, name in enumerate(self.vnorms):
            rf.reduce_mean(self.vnorms[name], mode="mean", out=per_layer[i:i + 1])
        rf.reduce_mean(per_layer, mode="mean", out=self.terms[2:3])
        lw = torch.tensor([3.0 * x_ for x_ in self.loss_weights], dtype=torch.float32, device=dev)
        weighted = rf.affine_rows(self.terms.view(3, 1), rf.hold(lw))
        self.loss = rf.reduce_mean(weighted.view(-1), mode="mean")

        def seed():
            npix = self.noise_pred.numel()
            ca = rb.hold(torch.full((1,), 2.0 * self.loss_weights[0] * self.grad_scale / npix, dtype=torch.float32, device=dev))
            cb = rb.hold(torch.full((1,), -2.0 * self.loss_weights[0] * self.grad_scale / npix, dtype=torch.float32, device=dev))
            seed_pred(rb.affine_rows(self.noise_pred.view(1, -1), ca, self.noise.view(1, -1), cb).view(self.noise_pred.shape))
        tp.back.append(seed)
        tp.build_backward()


    # ------------------------------------------------------------------ the identity-loss branch (train.py:521-535)
    _PASS_FIELDS = ("tape", "B", "E", "embs", "ids", "pidx", "pidx32", "timesteps", "t_state", "fusion_tab", "fusion_names", "vnorms",
                    "site_base", "vnorm_weight", "fusion_rng")

    def _enter(self, **kw):
        for k, v in kw.items():
            assert k in self._PASS_FIELDS, k
            setattr(self, k, v)

    def _snapshot(self):
        return {k: getattr(self, k) for k in self._PASS_FIELDS}

    def _build_face(self, ns, weight, guidance, T, noise_scheduler, fusion_seed, clip_tokens, clip_dim, image_size):
        """run_inference(sliced_batch, ..., guidance_scale, timesteps=T, token_index=0, from_noised_image=True, training_mode=True)
        (infer.py:7-123) + face_loss(pixel_values, gen_images, normalize=False) (loss.py:64-78) for ``ns`` samples:

            A  conditioning WITH gradient: text_adapter / image_adapter token 0, injected text encoder         (one recorder)
            L  one denoising step without gradient: uncond + cond UNet forward (one stacked pass), CFG + DPM-Solver++ update   (replayed T - 1 times)
            B  the last step with gradient: both forwards (per-layer fusion drawn on the device), the update,
               latents / scaling_factor -> VAE decode -> clamp -> ArcFace loss                                   (one recorder)
            backward of B and A in one plan.

        All UNet passes share the master weights (re-packed per step); the passes of L run the training-mode LoRA path (dropout active,
        like peft under set_cross_attention_layers_to_train) with the (1, 1) fusion of the no_grad rule."""
        import numpy as np
        from .attention_processor import PhotoVerseAttnProcessor2_0
        from .ops import Recorder
        from .scheduler import DPMSolverMultistepScheduler
        from .unet import Attention
        from .vae_train import decode_on_tape
        u, dev = self.unet, self.unet.device
        cfg = u.config
        h, w, seq = self.H, self.W, self.S_len
        main = self._snapshot()
        f = self.face = SimpleNamespace(ns=ns, weight=weight, guidance=guidance, T=T)
        sch = DPMSolverMultistepScheduler.from_config(noise_scheduler.config) if noise_scheduler is not None else DPMSolverMultistepScheduler()
        sch.set_timesteps(T)
        f.scheduler = sch
        coef_host = sch.coefficient_table()
        ft = f.tape = Tape(dev, self.grad_scale)
        hold = ft.rf.hold
        z32 = lambda *shape: hold(torch.zeros(shape, dtype=torch.float32, device=dev))
        f.lat, f.x0_prev = z32(ns, cfg.in_channels, h, w), z32(ns, cfg.in_channels, h, w)
        f.state = hold(torch.tensor([0, T, 0, 0], dtype=torch.int32, device=dev))
        f.coef = hold(coef_host.to(dev))
        f.ts = hold(sch.timesteps.to(dev, torch.float32))
        f.t_last = hold(torch.full((ns,), float(sch.timesteps[-1]), dtype=torch.float32, device=dev))
        f.embs_c = [hold(torch.zeros((ns, clip_tokens, clip_dim), dtype=torch.float16, device=dev))]
        f.embs_u = [hold(torch.zeros((ns, clip_tokens, clip_dim), dtype=torch.float16, device=dev))]
        f.ids = hold(torch.zeros((ns, seq), dtype=torch.int64, device=dev))
        f.pidx = hold(torch.zeros((ns,), dtype=torch.int64, device=dev))
        f.pidx32 = hold(torch.zeros((ns,), dtype=torch.int32, device=dev))
        f.text_u = hold(torch.zeros((ns * seq, cfg.cross_attention_dim), dtype=torch.float16, device=dev))
        f.real = z32(ns, 3, image_size, image_size)
        key = np.array([(fusion_seed + 0x51ED) & 0xFFFFFFFF, 0x2545F491, 0, 0], dtype=np.uint32).view(np.int32)
        rng = hold(torch.from_numpy(key.copy()).to(dev))
        procs = [m.processor for _, m in u.named_modules() if isinstance(m, Attention) and isinstance(m.processor, PhotoVerseAttnProcessor2_0)]
        nl = len(procs)

        # ---- A: conditioning with gradient (infer.py:86-96, token_index = 0)
        f.rec_cond = ft.rf
        self._enter(tape=ft, B=ns, E=1, embs=f.embs_c, ids=f.ids, pidx=f.pidx, pidx32=f.pidx32, timesteps=f.t_last, t_state=None, fusion_tab=None,
                    fusion_names=[], vnorms={}, site_base=1000, vnorm_weight=0.0, fusion_rng=rng)
        concept = self._adapter(self.text_adapter, f.embs_c)
        text_c = self._text(concept)
        ip_c = self._adapter(self.image_adapter, f.embs_c)
        ip_u = self._adapter(self.image_adapter, f.embs_u)
        text_u = Var(f.text_u, False)

        # ---- L: one denoising step without gradient, replayed T - 1 times (infer.py:98-119 under set_grad_enabled(False))
        # Without gradient both forwards use the (1, 1) fusion, so the uncond and the cond forward are ONE pass over the stacked batch
        # [uncond; cond] (same weights, per-sample conditioning): 2 ns rows fill the chip better than two passes of ns.
        tl = f.loop_tape = Tape(dev, 1.0)
        lrf = tl.rf
        one = lrf.hold(torch.ones((1,), dtype=torch.float32, device=dev))
        lat2 = lrf.hold(torch.zeros((2 * ns, cfg.in_channels, h, w), dtype=torch.float32, device=dev))
        text2 = lrf.hold(torch.zeros((2 * ns * seq, cfg.cross_attention_dim), dtype=torch.float16, device=dev))
        ip2 = lrf.hold(torch.zeros((2 * ns, cfg.cross_attention_dim), dtype=torch.float16, device=dev))
        zt = lrf.hold(torch.zeros((ns * seq, cfg.cross_attention_dim), dtype=torch.float16, device=dev))
        f.rec_stack = Recorder(dev)                       # once per iteration, after the conditioning: stack the two conditionings
        f.rec_stack.add_rows(f.text_u, zt, out=text2[:ns * seq])
        f.rec_stack.add_rows(text_c.t, zt, out=text2[ns * seq:])
        f.rec_stack.add_rows(ip_u.t, zt[:ns], out=ip2[:ns])
        f.rec_stack.add_rows(ip_c.t, zt[:ns], out=ip2[ns:])
        for half in (lat2[:ns], lat2[ns:]):               # every step: both halves read the current latents
            lrf.affine_rows(f.lat.view(1, -1), one, out=half.view(1, -1))
        # every replayed pass draws FRESH LoRA-dropout masks, as peft does on every forward: the masks are keyed on rng[2], which only the
        # fusion draws of section B advanced - one more counter bump per pass (the no-grad passes have no backward that would re-derive them)
        lrf.step_advance(rng[2:3])
        self._enter(tape=tl, B=2 * ns, t_state=(f.ts, f.state), fusion_names=[], site_base=2000)
        eps2, _ = self._unet_pass(lat2, Var(text2, False), Var(ip2, False))
        lrf.cfg_dpm_step(eps2[:ns], eps2[ns:], f.lat, f.x0_prev, f.coef, f.state, guidance)
        lrf.step_advance(f.state)
        tl.back = []
        self._enter(B=ns)

        # ---- B: the last step, in grad mode (infer.py:99), decode, loss
        f.rec_last = ft.rf = Recorder(dev)
        rf = ft.rf
        p0 = procs[0]
        f.fusion_tabs, f.fusion_forced = [], []
        for _ in range(2):
            tab = rf.hold(torch.ones((nl, 2), dtype=torch.float32, device=dev))
            forced = rf.hold(torch.full((nl,), -1.0, dtype=torch.float32, device=dev))
            rf.fusion_draw(None, rng, forced, tab, n_layers=nl, rule1=p0.fusion_rule1, rule2=p0.fusion_rule2, scale=float(p0.scale[0]), only_last_step=False)
            f.fusion_tabs.append(tab)
            f.fusion_forced.append(forced)
        self._enter(tape=ft, t_state=None, fusion_tab=f.fusion_tabs[0], fusion_names=[], site_base=3000, vnorms={})
        eps_u, seed_u = self._unet_pass(f.lat, text_u, ip_u)
        self._enter(fusion_tab=f.fusion_tabs[1], fusion_names=[], site_base=3500, vnorms={})
        eps_c, seed_c = self._unet_pass(f.lat, text_c, ip_c)
        f.fusion_names = list(self.fusion_names)
        rf.cfg_dpm_step(eps_u, eps_c, f.lat, f.x0_prev, f.coef, f.state, guidance)            # state[0] = T - 1 here: the final, first-order row
        sf = float(self.vae.config.scaling_factor)
        z = rf.affine_rows(f.lat.view(1, -1), rf.hold(torch.full((1,), 1.0 / sf, dtype=torch.float32, device=dev))).view(ns, cfg.in_channels, h, w)
        row = coef_host[T - 1]
        d_eps = float(row[3] * row[1])                     # d latents_final / d eps = c0 * cb   (x' = cx x + c0 (ca x + cb eps))
        rb = ft.rb

        def seeds():
            dz = dec.dz.g
            if dz is None:
                return
            cc = rb.hold(torch.full((1,), guidance * d_eps / sf, dtype=torch.float32, device=dev))
            seed_c(rb.affine_rows(dz.reshape(1, -1), cc).view(dz.shape))
            if guidance != 1.0:
                cu = rb.hold(torch.full((1,), (1.0 - guidance) * d_eps / sf, dtype=torch.float32, device=dev))
                seed_u(rb.affine_rows(dz.reshape(1, -1), cu).view(dz.shape))
        ft.back.append(seeds)
        dec = decode_on_tape(ft, self.vae, z)
        fl = self.face_loss.attach(ft, f.real, dec.img, weight=weight, maximize=True, normalize=False, holder=dec.dimg)
        f.loss, f.images = fl.loss, dec.img
        ft.build_backward()
        self._enter(**main)

    @torch.no_grad()
    def _run_face(self, fi):
        """``fi``: pixel_values (ns, 3, H, W) fp32 - the real images; start_latents (ns, 4, h, w) - vae-encoded, scaled and noised to the
        first timestep of the inference schedule (infer.py:62-68); image_embeddings / uncond_image_embeddings: token-0 CLIP hidden states
        (ns, 257, 1024) of the image and of the zero image (:76-84); text_input_ids / placeholder_idx of "a photo of *"; uncond_input_ids."""
        f = self.face
        f.real.copy_(fi["pixel_values"])
        f.lat.copy_(fi["start_latents"])
        f.x0_prev.zero_()
        f.state.copy_(torch.tensor([0, f.T, 0, 0], dtype=torch.int32))
        f.embs_c[0].copy_(fi["image_embeddings"])
        f.embs_u[0].copy_(fi["uncond_image_embeddings"])
        f.ids.copy_(fi["text_input_ids"].view(f.ns, -1))
        f.pidx.copy_(fi["placeholder_idx"].view(-1))
        f.pidx32.copy_(fi["placeholder_idx"].view(-1))
        f.text_u.copy_(self.text_encoder({"text_input_ids": fi["uncond_input_ids"].to(f.ids.device)})[0].reshape(f.text_u.shape))
        for forced, src in zip(f.fusion_forced, fi.get("forced_fusion") or (None, None)):
            if src is None:
                forced.fill_(-1.0)
            else:
                forced.copy_(torch.as_tensor(src, dtype=torch.float32))
        f.tape.load_weights()
        f.loop_tape.load_weights()

        def replay():
            f.rec_cond.run()
            f.rec_stack.run()
            for _ in range(f.T - 1):
                f.loop_tape.rf.run()
            f.rec_last.run()
            f.tape.rb.run()
        if getattr(f, "graph", None) is not None:
            f.graph.replay()
        elif self.use_graph and getattr(f, "warm", False):
            torch.cuda.synchronize()                      # second iteration on: the whole branch (~13k launches at T = 10) as one HIP graph
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                replay()
            f.graph = gr
            f.state.copy_(torch.tensor([0, f.T, 0, 0], dtype=torch.int32))     # capture does not execute: the counter is still at 0
            gr.replay()
        else:
            replay()
            f.warm = True

    # ------------------------------------------------------------------ one iteration
    def trainable_parameters(self) -> Dict[str, List[torch.nn.Parameter]]:
        """The three clip groups of train.py:538-541 (text_adapter, image_adapter, unet)."""
        from .attention_processor import PhotoVerseAttnProcessor2_0
        un = []
        for proc in self.unet.attn_processors.values():
            if isinstance(proc, PhotoVerseAttnProcessor2_0):
                un += [proc.to_k_ip[0].weight, proc.to_v_ip[0].weight]
        for m in self.unet.modules():
            if isinstance(m, LoRALinear):
                un += [m.lora_A["default"].weight, m.lora_B["default"].weight]
        return {"text_adapter": list(self.text_adapter.parameters()), "image_adapter": list(self.image_adapter.parameters()), "unet": un}

    @torch.no_grad()
    def step(self, *, noisy_latents, noise, timesteps, text_input_ids, placeholder_idx, image_embeddings, forced_fusion=None, face_inputs=None,
             accumulate: bool = False):
        """Fill the static inputs, replay forward + backward, set ``.grad`` (= gradient x ``grad_scale``).  Returns the loss terms.
        ``face_inputs`` (see ``_run_face``): required when the step was built with a face loss.  ``accumulate``: ADD this micro-batch's
        gradients to the ones of the previous calls (``accelerator.accumulate``, train.py:464) in persistent fp32 accumulators; the first
        micro-batch of an update is a call with ``accumulate=False``; divide by the count through the optimizer's ``grad_scale``."""
        if (self.face is None) != (face_inputs is None):
            raise ValueError("face_inputs must be given exactly when the TrainStep was built with face_loss")
        self.noisy.copy_(noisy_latents)
        self.noise.copy_(noise)
        self.timesteps.copy_(timesteps.to(self.noisy.device, torch.float32))
        self.ids.copy_(text_input_ids.view(self.B, -1))
        self.pidx.copy_(placeholder_idx.view(-1))
        self.pidx32.copy_(placeholder_idx.view(-1))
        for dst, src in zip(self.embs, image_embeddings):
            dst.copy_(src)
        if forced_fusion is not None:
            self.fusion_forced.copy_(torch.as_tensor(forced_fusion, dtype=torch.float32))
        else:
            self.fusion_forced.fill_(-1.0)
        self.tape.load_weights()
        if self.graph is not None:
            self.graph.replay()
        elif self.use_graph and self._warm:
            # second iteration on: forward + backward plans as ONE HIP graph (launches enqueue on the capturing stream)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                self.tape.rf.run()
                self.tape.rb.run()
            self.graph = gr
            gr.replay()
        else:
            self.tape.rf.run()
            self.tape.rb.run()
            self._warm = True
        if self.face is not None:
            self._run_face(face_inputs)
        seen = set()
        if accumulate:                                          # keep the running sums: the plan buffers are rewritten by the next replay
            if not hasattr(self, "_accum"):
                raise RuntimeError("accumulate=True needs a preceding step(accumulate=False) of the same update")
            seen = set(self._accum)                              # every parameter already has a gradient: _give adds
            for pid, acc in self._accum.items():
                self._accum_params[pid].grad = acc
        for param, buf in self.pgrads:                          # a parameter used by several passes has several buffers: summed here
            self._give(param, buf() if callable(buf) else buf.view(param.shape), seen)
        for mod, dW in self.lora_pending:                       # rank-r factor gradients from the merged-weight gradient
            A, Bm = mod.lora_A["default"].weight, mod.lora_B["default"].weight
            self._give(A, mod.scaling * (Bm.detach().t() @ dW), seen)
            self._give(Bm, mod.scaling * (dW @ A.detach().t()), seen)
        # gradients that may be needed after the next replay (accumulation) live in buffers of their own
        self._accum, self._accum_params = {}, {}
        for ps in self.trainable_parameters().values():
            for prm in ps:
                if prm.grad is not None:
                    self._accum[id(prm)], self._accum_params[id(prm)] = prm.grad, prm
        out = {"loss": self.loss, "diffusion_loss": self.terms[0:1], "concept_text_loss": self.terms[1:2],
               "cross_attn_visual_loss": self.terms[2:3], "noise_pred": self.noise_pred, "fusion_table": self.fusion_tab}
        if self.face is not None:
            # loss (train.py:535) = the three terms + 0.01 * floss: one more row-affine launch
            rec = _ops.Recorder(self.noisy.device)
            wv = rec.hold(torch.full((1,), self.face.weight, dtype=torch.float32, device=self.noisy.device))
            one = rec.hold(torch.ones((1,), dtype=torch.float32, device=self.noisy.device))
            out["loss"] = rec.affine_rows(self.loss.view(1, 1), one, self.face.loss.view(1, 1), wv).view(1)
            rec.run()
            out["face_loss"], out["face_images"] = self.face.loss, self.face.images
        return out

    def _give(self, param, g, seen):
        if id(param) not in seen:
            seen.add(id(param))
            param.grad = g
            return
        rec = _ops.Recorder(g.device)
        one = rec.hold(torch.ones((1,), dtype=torch.float32, device=g.device))
        out = rec.affine_rows(param.grad.reshape(1, -1).contiguous(), one, g.reshape(1, -1).contiguous(), one)
        rec.run()
        param.grad = out.view(param.shape)


@torch.no_grad()
def training_iteration(step: TrainStep, optimizer, batch, tokenizer, image_encoder, vae, noise_scheduler, device, image_encoder_layers_idx,
                       extra_num_tokens: int, *, generator: Optional[torch.Generator] = None, max_grad_norm: float = 1.0, micro_step: int = 0,
                       accumulation_steps: int = 1, reducer: Optional["GradientReducer"] = None):
    """The body of the reference's training loop (``train.py:464-549``) on top of ``TrainStep``: the frozen, gradient-free part with this
    package's inference modules - VAE encode + posterior sample (``:471-472``), noise / per-sample timesteps / add_noise (``:475-484``), CLIP
    image features (``:487-492``) and, with a face loss, the inputs of ``run_inference(sliced_batch, ...)`` (``:522-530``, ``infer.py:42-84``:
    the "a photo of *" prompt, a random subset of the batch, its latents noised to the first inference timestep, the zero-image features) -
    then the forward + backward plans and the clipped AdamW update (``:536-547``).  Returns the loss terms of the step.
    ``accumulation_steps`` > 1 (``--gradient_accumulation_steps``): micro-batch ``micro_step`` of that many; gradients are summed, the optimizer
    runs after the last one on their mean (accelerate scales each micro-batch's loss by 1 / steps).  ``reducer``: a ``GradientReducer`` over the
    trainable parameters when several ranks train data-parallel - the gradients are summed over the ranks once per optimizer step (not per
    micro-batch, like DDP under ``accelerator.accumulate``) and the optimizer takes their mean."""
    device = torch.device(device)
    pixel_values = batch["pixel_values"].to(device, dtype=torch.float32)
    pixel_values_clip = batch["pixel_values_clip"].to(device, dtype=torch.float32)
    bsz = pixel_values.shape[0]
    cpu_gen = dict(generator=generator) if generator is not None else {}
    rec = Recorder(device)
    dist = vae.encode(pixel_values).latent_dist
    eps = torch.randn(dist.mean.shape, **cpu_gen).to(device, torch.float32)
    sf = torch.full((bsz,), float(vae.config.scaling_factor), dtype=torch.float32, device=device)
    latents = rec.affine_rows(rec.posterior_sample(dist.parameters.contiguous(), eps.contiguous()), sf)
    rec.run()
    noise = torch.randn(latents.shape, **cpu_gen).to(device, torch.float32)
    n_train = noise_scheduler.config["num_train_timesteps"] if isinstance(noise_scheduler.config, dict) else noise_scheduler.config.num_train_timesteps
    timesteps = torch.randint(0, n_train, (bsz,), **cpu_gen).long()
    from .scheduler import DPMSolverMultistepScheduler
    sch = noise_scheduler if hasattr(noise_scheduler, "add_noise") else DPMSolverMultistepScheduler.from_config(noise_scheduler.config)
    noisy_latents = sch.add_noise(latents, noise, timesteps)
    feats = image_encoder(pixel_values_clip, output_hidden_states=True)
    image_embeddings = [feats[0]] + [feats[2][i] for i in image_encoder_layers_idx if i < len(feats[2])]
    assert len(image_embeddings) == extra_num_tokens + 1, "Entered indices are out of range for image_encoder layers."
    face_inputs = None
    if step.face is not None:
        f = step.face
        idx = torch.randperm(bsz, **cpu_gen)[:f.ns]                                              # random_batch_slicing (datasets/utils.py:223-234)
        text = "a photo of {}".format("*")                                                       # prepare_prompt (train.py:523)
        ids = tokenizer([text] * f.ns, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
        uids = tokenizer([""] * f.ns, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
        pidx = torch.full((f.ns, 1), text.split().index("*") + 1, dtype=torch.int64)
        sub = pixel_values[idx.to(device)]
        d2 = vae.encode(sub).latent_dist                                                         # infer.py:62-65 (a fresh posterior sample)
        rec = Recorder(device)
        e2 = torch.randn(d2.mean.shape, **cpu_gen).to(device, torch.float32)
        lat = rec.affine_rows(rec.posterior_sample(d2.parameters.contiguous(), e2.contiguous()), sf[:f.ns].contiguous())
        rec.run()
        n2 = torch.randn(lat.shape, **cpu_gen).to(device, torch.float32)
        start = f.scheduler.add_noise(lat, n2, f.scheduler.timesteps[:1].repeat(f.ns)) * f.scheduler.init_noise_sigma
        ufeats = image_encoder(torch.zeros_like(pixel_values_clip[:f.ns]), output_hidden_states=True)
        face_inputs = dict(pixel_values=sub, start_latents=start, image_embeddings=image_embeddings[0][idx.to(device)],
                           uncond_image_embeddings=ufeats[0], text_input_ids=ids.to(device), placeholder_idx=pidx.to(device),
                           uncond_input_ids=uids.to(device))
    out = step.step(noisy_latents=noisy_latents, noise=noise, timesteps=timesteps, text_input_ids=batch["text_input_ids"].to(device),
                    placeholder_idx=batch["concept_placeholder_idx"].to(device), image_embeddings=image_embeddings, face_inputs=face_inputs,
                    accumulate=micro_step > 0)
    if micro_step + 1 < accumulation_steps:
        if micro_step == 0 and accumulation_steps > 1:
            _detach_grads(step)                              # the first micro-batch's gradients alias plan buffers the next replay rewrites
        return out
    groups = step.trainable_parameters()
    world = reducer() if reducer is not None else 1
    out["grad_norms"] = optimizer.step(clip_groups=list(groups.values()), max_norm=max_grad_norm,
                                       grad_scale=step.grad_scale * accumulation_steps * world)
    optimizer.zero_grad()
    return out


class GradientReducer:
    """Data-parallel training: the reference runs its loop under accelerate (``train.py:299-305``, ``accelerator.prepare`` ``:398-400``), i.e.
    every rank steps its own shard of the batch and DDP averages the gradients of the trainable parameters before ``optimizer.step()``.

    Here: ONE flat fp32 bucket for all trainable gradients (≈ 220 tensors, tens of MB with the reference's LoRA rank - far below the size at
    which a ring all-reduce over xGMI becomes bandwidth-bound, so splitting it into buckets would only add launches) and ONE
    ``all_reduce(SUM)`` per optimizer step (RCCL when the process group's backend is ``nccl``).  The division by the world size is not a
    pass over the bucket: the caller multiplies the optimizer's ``grad_scale`` by the returned world size.  After the call every
    ``param.grad`` is a view of the bucket, so the multi-tensor optimizer keeps seeing the same addresses from step to step.  The backward
    plan has fully run when the bucket is filled (the plan is one HIP graph): there is no backward left to overlap the collective with."""

    def __init__(self, params, group=None, force: bool = False):
        """``force``: run the collective even in a one-rank group (exercises the RCCL path on a single GPU)."""
        import torch.distributed as dist
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.force = bool(force)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()

    @torch.no_grad()
    def __call__(self) -> int:
        """Sum the gradients over the ranks in place; returns the number of ranks the sum ran over (1: nothing to do)."""
        if self.world == 1 and not self.force:
            return 1
        import torch.distributed as dist
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if len(have) != len(self.params):
            self.flat.zero_()                                # a parameter outside this step's plan contributes nothing on this rank
        torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            host = self.flat.cpu()                           # gloo (CPU rendezvous, tests on one device): through host memory
            dist.all_reduce(host, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, group=self.group)
        for p, v in have:
            p.grad = v
        return self.world


def _detach_grads(step: TrainStep):
    """Copy every gradient out of the plan's buffers (fp32 row-affine kernel as the copy) so that it survives the next replay."""
    rec = Recorder(step.noisy.device)
    one = rec.hold(torch.ones((1,), dtype=torch.float32, device=step.noisy.device))
    new = {}
    for pid, g in step._accum.items():
        new[pid] = rec.affine_rows(g.reshape(1, -1).contiguous(), one).view(g.shape)
    rec.run()
    for pid, g in new.items():
        step._accum[pid] = g
        step._accum_params[pid].grad = g
