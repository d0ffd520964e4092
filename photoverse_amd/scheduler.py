"""DPM-Solver++(2M) tables for the device-side step kernel.

The reference rebuilds ``DPMSolverMultistepScheduler.from_config(DDPMScheduler.config)`` per call and drives it from
Python (``/root/reference/models/infer.py:39-40,70,100,119``).  All of its per-step arithmetic that is not elementwise
on the latents is scalar, so it is evaluated once on the host into a coefficient table and the elementwise part runs in
``pv_cfg_dpm_step`` (one launch per step, indexed by a device-resident step counter -> graph-replayable).

Config restated from the SD-v1.5 scheduler the reference loads (``modeling_utils.py:60``): ``scaled_linear`` betas
0.00085 -> 0.012, 1000 training steps, ``steps_offset=1``, ``timestep_spacing="leading"``, epsilon prediction;
DPM-Solver defaults ``solver_order=2``, ``dpmsolver++``, ``midpoint``, ``lower_order_final``, ``final_sigmas_type="zero"``.
[EXT diffusers 0.27.2 - restated from the published algorithm, parity unpinned.]
"""
from __future__ import annotations

import numpy as np
import torch


class DPMSolverMultistepScheduler:
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule="scaled_linear", steps_offset=steps_offset, timestep_spacing="leading",
                           solver_order=2, algorithm_type="dpmsolver++", solver_type="midpoint", prediction_type="epsilon")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.timesteps = None

    @classmethod
    def from_config(cls, config):
        cfg = config if isinstance(config, dict) else getattr(config, "__dict__", {})
        return cls(cfg.get("num_train_timesteps", 1000), cfg.get("beta_start", 0.00085), cfg.get("beta_end", 0.012),
                   cfg.get("steps_offset", 1))

    def set_timesteps(self, n: int):
        T = self.config["num_train_timesteps"]
        step_ratio = T // (n + 1)
        ts = (np.arange(0, n + 1) * step_ratio).round()[::-1][:-1].copy().astype(np.int64) + self.config["steps_offset"]
        sig_all = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        sig = np.interp(ts, np.arange(len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n

    def scale_model_input(self, sample, t):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(acp[t]) x0 + sqrt(1 - acp[t]) noise  (infer.py:65, train.py:484); ``timesteps``: int64 [B]."""
        acp = torch.from_numpy(self.alphas_cumprod)[timesteps.cpu().long()].to(torch.float64)          # host table lookup (per-sample scalars)
        ca, cb = acp.sqrt().float(), (1 - acp).sqrt().float()
        if original_samples.is_cuda and original_samples.dtype == torch.float32:
            from .ops import Recorder
            rec = Recorder(original_samples.device)
            out = rec.affine_rows(original_samples.contiguous(), ca.to(original_samples.device), noise.contiguous().to(original_samples.device),
                                  cb.to(original_samples.device))
            rec.run()
            return out
        shape = (-1, *([1] * (original_samples.dim() - 1)))
        return ca.view(shape).to(original_samples) * original_samples + cb.view(shape).to(original_samples) * noise

    def coefficient_table(self) -> torch.Tensor:
        """float32 [n, 8] rows {ca, cb, cx, c0, c1, 0, 0, 0} with, per step i (s = current, t = next sigma):
            x0     = ca*x + cb*eps            ca = 1/alpha_s, cb = -sigma_s/alpha_s
            x_next = cx*x + c0*x0 + c1*x0_prev
        first order (first step, last step):  cx = sig_t/sig_s, c0 = -alpha_t*(exp(-h)-1), c1 = 0
        second order (midpoint):              c0 = -c*(1 + 1/(2 r0)), c1 = c/(2 r0),  c = alpha_t*(exp(-h)-1)
        """
        n = self.num_inference_steps
        sig = self.sigmas.astype(np.float64)
        alpha = 1.0 / np.sqrt(sig * sig + 1.0)
        sigma = sig * alpha
        with np.errstate(divide="ignore"):
            lam = np.log(alpha) - np.log(sigma)
        tab = np.zeros((n, 8), dtype=np.float64)
        for i in range(n):
            a_s, s_s, a_t, s_t = alpha[i], sigma[i], alpha[i + 1], sigma[i + 1]
            h = lam[i + 1] - lam[i]
            c = a_t * (np.exp(-h) - 1.0)
            # first step (no history) and last step (final sigma 0 -> lower_order_final) are 1st order; with
            # solver_order 2 every other step is the 2nd-order multistep update
            first = i == 0 or i == n - 1
            tab[i, 0], tab[i, 1], tab[i, 2] = 1.0 / a_s, -s_s / a_s, s_t / s_s
            if first:
                tab[i, 3], tab[i, 4] = -c, 0.0
            else:
                r0 = (lam[i] - lam[i - 1]) / h
                tab[i, 3], tab[i, 4] = -c * (1.0 + 0.5 / r0), 0.5 * c / r0
        return torch.from_numpy(tab.astype(np.float32))


class DDIMScheduler(DPMSolverMultistepScheduler):
    """Deterministic DDIM (eta = 0) on the same beta schedule / timestep grid, expressed in the SAME coefficient-table form
    (``c1 = 0``), so the loop kernel and the captured graph are unchanged.  The reference's sampler is DPM-Solver++
    (``infer.py:39-40``); BASELINE.json's metric text says "DDIM", so both are selectable (``DenoiseLoop(scheduler=...)``).

        x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t)
        x' = sqrt(a_p) x0 + sqrt(1-a_p) eps,   eps = (x - sqrt(a_t) x0) / sqrt(1-a_t)
           = [sqrt(1-a_p)/sqrt(1-a_t)] x + [sqrt(a_p) - sqrt(a_t) sqrt(1-a_p)/sqrt(1-a_t)] x0
    with a_t = alphas_cumprod[t], a_p = alphas_cumprod[t - T/n] (1.0 past the last step, diffusers ``set_alpha_to_one``
    is False for SD-v1.5 -> alphas_cumprod[0]).
    """

    def coefficient_table(self) -> torch.Tensor:
        n = self.num_inference_steps
        T = self.config["num_train_timesteps"]
        ratio = T // n
        ts = self.timesteps.numpy()
        acp = self.alphas_cumprod.astype(np.float64)
        tab = np.zeros((n, 8), dtype=np.float64)
        for i, t in enumerate(ts):
            a_t = acp[t]
            prev = t - ratio
            a_p = acp[prev] if prev >= 0 else acp[0]
            r = np.sqrt(1 - a_p) / np.sqrt(1 - a_t)
            tab[i, 0], tab[i, 1] = 1.0 / np.sqrt(a_t), -np.sqrt(1 - a_t) / np.sqrt(a_t)
            tab[i, 2], tab[i, 3], tab[i, 4] = r, np.sqrt(a_p) - np.sqrt(a_t) * r, 0.0
        return torch.from_numpy(tab.astype(np.float32))

    def set_timesteps(self, n: int):
        # diffusers DDIMScheduler "leading" spacing: arange(n) * (T // n) reversed, + steps_offset
        T = self.config["num_train_timesteps"]
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64) + self.config["steps_offset"]
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self.sigmas = None
