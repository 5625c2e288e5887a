"""Flow-matching DPM-Solver++ scheduler with the reference's interface, latent updates on the GPU via HIP kernels.

Mirrors /root/reference/models/wan/utils/fm_solvers.py (get_sampling_sigmas :24-28, retrieve_timesteps :31-68,
FlowDPMSolverMultistepScheduler :71-859) for the configuration WanTI2V instantiates when `sample_solver='dpm++'`
(models/wan/textimage2video.py:343-351): solver_order 2 (or 1), dpmsolver++, midpoint, flow_prediction, final sigma zero,
no thresholding. As in `fm_solvers_unipc.py` the scalar coefficient algebra stays on the host in 0-dim fp32 tensors exactly
like the reference; the latent-sized update is one fused HBM pass (csrc/sampler.hip: uv_cfg_convert, uv_dpmpp_update) with the
reference's rounding sequence, so given identical model outputs the trajectory is bit-identical.
"""
import inspect

import numpy as np
import torch

from .. import _lib
from .fm_solvers_unipc import SchedulerOutput


def get_sampling_sigmas(sampling_steps, shift):
    """fm_solvers.py:24-28."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """fm_solvers.py:31-68: hands a custom `timesteps` or `sigmas` schedule (or a plain step count) to
    `scheduler.set_timesteps` and returns `(scheduler.timesteps, number of steps)`. ValueError if both custom schedules are
    given, or if the scheduler's `set_timesteps` has no such parameter."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("pass either `timesteps` or `sigmas`, not both")
    accepted = inspect.signature(scheduler.set_timesteps).parameters
    custom = {"timesteps": timesteps, "sigmas": sigmas}
    for name, value in custom.items():
        if value is None:
            continue
        if name not in accepted:
            raise ValueError(f"{type(scheduler).__name__}.set_timesteps takes no `{name}` argument: this scheduler cannot run a "
                             f"custom {name} schedule")
        scheduler.set_timesteps(device=device, **{name: value}, **kwargs)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


class FlowDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: float = 1.0, use_dynamic_shifting=False, thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0, algorithm_type: str = "dpmsolver++",
                 solver_type: str = "midpoint", lower_order_final: bool = True, euler_at_final: bool = False,
                 final_sigmas_type: str = "zero", lambda_min_clipped: float = -float("inf"), variance_type=None,
                 invert_sigmas: bool = False):
        if (solver_order not in (1, 2) or prediction_type != "flow_prediction" or use_dynamic_shifting or thresholding
                or algorithm_type != "dpmsolver++" or solver_type != "midpoint" or final_sigmas_type != "zero"):
            raise NotImplementedError("only WanTI2V's DPM-Solver++ setting is built: order <= 2, dpmsolver++, midpoint, "
                                      "flow_prediction, final sigma zero")
        self.config = type("Config", (), dict(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                                              shift=shift, lower_order_final=lower_order_final, euler_at_final=euler_at_final,
                                              solver_type=solver_type, algorithm_type=algorithm_type,
                                              prediction_type=prediction_type, final_sigmas_type=final_sigmas_type))()
        self.num_inference_steps = None
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = sigmas
        self.timesteps = sigmas * num_train_timesteps
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        """fm_solvers.py:228-291."""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.config.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.config.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self._timesteps_host = self.timesteps.cpu().tolist()
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        self._begin_index = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        t = int(timestep)
        idx = [i for i, v in enumerate(self._timesteps_host) if v == t]
        self._step_index = idx[1] if len(idx) > 1 else idx[0]

    # ---- host coefficient algebra (0-dim fp32 tensors; the expressions of :466-476 and :541-562) ------------------
    def _coeffs(self, order):
        si = self._step_index
        sigma_t, sigma_s0 = self.sigmas[si + 1], self.sigmas[si]
        alpha_t = 1 - sigma_t
        lam = lambda s: torch.log(1 - s) - torch.log(s)
        lambda_t, lambda_s0 = lam(sigma_t), lam(sigma_s0)
        h = lambda_t - lambda_s0
        inv_r0 = 1.0
        if order == 2:
            h_0 = lambda_s0 - lam(self.sigmas[si - 1])
            inv_r0 = (1.0 / (h_0 / h)).item()
        return (sigma_t / sigma_s0).item(), (alpha_t * (torch.exp(-h) - 1.0)).item(), inv_r0

    def _advance(self, x0, sample):
        """Everything in step() after convert_model_output (:779-800)."""
        n = len(self._timesteps_host)
        lower_order_final = self._step_index == n - 1 and (
            self.config.euler_at_final or (self.config.lower_order_final and n < 15) or self.config.final_sigmas_type == "zero")
        for i in range(self.config.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        order = 1 if (self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final) else 2
        r, c, inv_r0 = self._coeffs(order)
        out = torch.empty_like(sample)
        m1 = self.model_outputs[-2] if order == 2 else None
        _lib.call("uv_dpmpp_update", _lib.ptr(sample), _lib.ptr(x0), _lib.ptr(m1), _lib.ptr(out), r, c, inv_r0, order,
                  out.numel(), _lib.stream_ptr())
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return out

    def _check(self, *tensors):
        for t in tensors:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.UnividHipError("DPM-Solver++: latents must be contiguous fp32 GPU tensors")

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict: bool = True):
        """fm_solvers.py:708-800."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        self._check(model_output, sample)
        if self._step_index is None:
            self._init_step_index(timestep)
        x0 = torch.empty_like(sample)
        sigma = self.sigmas[self._step_index].item()
        # gs = 0 makes the CFG stage the identity: x0 = sample - sigma * model_output (:393-394)
        _lib.call("uv_cfg_convert", _lib.ptr(model_output), _lib.ptr(model_output), _lib.ptr(sample), 0.0, sigma, None,
                  _lib.ptr(x0), x0.numel(), _lib.stream_ptr())
        prev = self._advance(x0, sample)
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def step_cfg(self, cond, uncond, guide_scale, timestep, sample, want_noise_pred=False):
        """CFG combine (textimage2video.py:385) fused with convert_model_output, then the usual update."""
        self._check(cond, uncond, sample)
        if self._step_index is None:
            self._init_step_index(timestep)
        x0 = torch.empty_like(sample)
        npred = torch.empty_like(sample) if want_noise_pred else None
        sigma = self.sigmas[self._step_index].item()
        _lib.call("uv_cfg_convert", _lib.ptr(cond), _lib.ptr(uncond), _lib.ptr(sample), float(guide_scale), sigma,
                  _lib.ptr(npred), _lib.ptr(x0), x0.numel(), _lib.stream_ptr())
        prev = self._advance(x0, sample)
        return (prev, npred) if want_noise_pred else prev
