"""Oracle: DPM-Solver++(2M) as the reference uses it.  TEST INFRASTRUCTURE.
**PARITY UNPINNED** against the library ([EXT] ``diffusers==0.27.2`` ``DPMSolverMultistepScheduler``, not installable);
the algorithm itself is checked against the closed-form probability-flow solution for Gaussian data
(``tests/test_oracle_pins.py::test_samplers_converge_to_the_exact_probability_flow_solution``).

The reference rebuilds the sampler on every call with
``DPMSolverMultistepScheduler.from_config(scheduler.config)`` where
``scheduler`` is the SD-v1.5 ``DDPMScheduler``
(``/root/reference/models/infer.py:39-40``, ``modeling_utils.py:60``), then calls
``set_timesteps`` (:40), ``init_noise_sigma`` (:70), ``scale_model_input``
(:100) and ``step`` (:119).  Restated from the published algorithm
(Lu et al., DPM-Solver++, multistep 2M, midpoint) with the config the
reference ends up with: betas ``scaled_linear`` 0.00085->0.012 over 1000
training steps, ``steps_offset=1``, ``timestep_spacing="leading"`` (inherited
from the SD-v1.5 scheduler config), ``solver_order=2``,
``algorithm_type="dpmsolver++"``, ``solver_type="midpoint"``,
``lower_order_final=True``, ``final_sigmas_type="zero"``, epsilon prediction.
"""
import numpy as np
import torch


class DPMSolverMultistepRef:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        step_ratio = self.num_train_timesteps // (n + 1)
        ts = (np.arange(0, n + 1) * step_ratio).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        sig_all = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        sig = np.interp(ts, np.arange(len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        self.step_index = 0

    def scale_model_input(self, sample, t):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / np.sqrt(sigma * sigma + 1.0)
        return alpha_t, sigma * alpha_t

    def step(self, model_output, t, sample):
        i = self.step_index
        n = len(self.timesteps)
        lower_order_final = i == n - 1        # final_sigmas_type == "zero"
        lower_order_second = (i == n - 2) and n < 15
        a_s, s_s = self._alpha_sigma(np.float64(self.sigmas[i]))
        x0 = (sample - float(s_s) * model_output) / float(a_s)     # epsilon -> data prediction
        self.model_outputs[0] = self.model_outputs[1]
        self.model_outputs[1] = x0
        a_t, s_t = self._alpha_sigma(np.float64(self.sigmas[i + 1]))
        with np.errstate(divide="ignore"):
            lam_t = np.log(a_t) - np.log(s_t)
            lam_s = np.log(a_s) - np.log(s_s)
        h = lam_t - lam_s
        if self.lower_order_nums < 1 or lower_order_final:
            prev = float(s_t / s_s) * sample - float(a_t * (np.exp(-h) - 1.0)) * x0
        else:
            a_s1, s_s1 = self._alpha_sigma(np.float64(self.sigmas[i - 1]))
            lam_s1 = np.log(a_s1) - np.log(s_s1)
            r0 = (lam_s - lam_s1) / h
            d0 = self.model_outputs[1]
            d1 = (1.0 / float(r0)) * (self.model_outputs[1] - self.model_outputs[0])
            c = float(a_t * (np.exp(-h) - 1.0))
            prev = float(s_t / s_s) * sample - c * d0 - 0.5 * c * d1
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev

    def add_noise(self, original_samples, noise, timesteps):
        """``infer.py:65`` / ``train.py:484``.  [EXT] diffusers 0.27.2 ``DPMSolverMultistepScheduler.add_noise``: the sigma of each
        timestep's position in the CURRENT schedule (``index_for_timestep``: the first match of a unique schedule), then
        ``alpha_t * x0 + sigma_t * noise`` with ``alpha_t = 1 / sqrt(sigma^2 + 1)``, ``sigma_t = sigma * alpha_t``."""
        sched = self.timesteps.tolist()
        idx = [sched.index(int(t)) for t in timesteps]
        sigma = torch.from_numpy(self.sigmas[idx]).to(original_samples.dtype)
        alpha_t = 1.0 / torch.sqrt(sigma * sigma + 1.0)
        sigma_t = sigma * alpha_t
        shape = (-1, *([1] * (original_samples.dim() - 1)))
        return alpha_t.view(shape) * original_samples + sigma_t.view(shape) * noise


class DDIMRef:
    """Stepwise DDIM (eta = 0, epsilon prediction, "leading" spacing, steps_offset 1, set_alpha_to_one False) - the sampler
    BASELINE.json's metric text names; [EXT diffusers DDIMScheduler], PARITY UNPINNED."""
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        self.T, self.steps_offset = num_train_timesteps, steps_offset

    def set_timesteps(self, n):
        self.ratio = self.T // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * self.ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset)

    def step(self, eps, t, x):
        t = int(t)
        a_t = self.alphas_cumprod[t]
        prev = t - self.ratio
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.alphas_cumprod[0]
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps
