"""Flow-matching UniPC scheduler with the reference's interface, latent updates on the GPU via HIP kernels.

Mirrors FlowUniPCMultistepScheduler (/root/reference/models/wan/utils/fm_solvers_unipc.py:22-803) for the
configuration WanTI2V instantiates (models/wan/textimage2video.py:336-342): solver_order 2, bh2, predict_x0,
flow_prediction, lower_order_final, final sigma zero. Scalar coefficient algebra stays on the host in fp32
0-dim tensors exactly like the reference (its sigmas are kept on the CPU "to avoid too much CPU/GPU
communication", :131,228); only the latent-sized AXPYs run on the device (csrc/sampler.hip), with the
reference's rounding sequence.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .. import _lib


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: float = 1.0, use_dynamic_shifting=False, thresholding: bool = False, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector=(),
                 final_sigmas_type: str = "zero"):
        if (solver_order != 2 or prediction_type != "flow_prediction" or use_dynamic_shifting or thresholding
                or not predict_x0 or solver_type != "bh2" or final_sigmas_type != "zero"):
            raise NotImplementedError("only UniVid's UniPC setting is built: order 2, bh2, predict_x0, flow_prediction")
        self.config = type("Config", (), dict(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                                              shift=shift, lower_order_final=lower_order_final,
                                              solver_type=solver_type, prediction_type=prediction_type))()
        self.predict_x0 = True
        self.num_inference_steps = None
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = sigmas.to("cpu")
        self.timesteps = sigmas * num_train_timesteps
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.model_outputs = [None] * solver_order
        self.timestep_list = [None] * solver_order
        self.lower_order_nums = 0
        self.disable_corrector = list(disable_corrector)
        self.last_sample = None
        self._step_index = None
        self._begin_index = None
        self.this_order = None

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        """fm_solvers_unipc.py:162-229."""
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
        self.last_sample = None
        self._step_index = None
        self._begin_index = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    # ---- host coefficient algebra (0-dim fp32 tensors, same expressions as :407-455 / :550-598) ---------------
    def _coeffs(self, i_t, i_s0, order, hist):
        sigma_t, sigma_s0 = self.sigmas[i_t], self.sigmas[i_s0]
        alpha_t, alpha_s0 = 1 - sigma_t, 1 - sigma_s0
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = []
        for si in hist:
            lam = torch.log(1 - self.sigmas[si]) - torch.log(self.sigmas[si])
            rks.append((lam - lambda_s0) / h)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        R, b = [], []
        rks_t = torch.tensor(rks + [1.0])
        fact = 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return dict(r=(sigma_t / sigma_s0).item(), c1=(alpha_t * h_phi_1).item(), c2=(alpha_t * B_h).item(),
                    rk=(rks[0].item() if rks else 1.0), R=torch.stack(R), b=torch.tensor(b))

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        t = int(timestep)
        idx = [i for i, v in enumerate(self._timesteps_host) if v == t]
        self._step_index = idx[1] if len(idx) > 1 else idx[0]

    # ---- device updates ---------------------------------------------------------------------------------------
    def _correct(self, x0, last_sample, order):
        si = self._step_index
        c = self._coeffs(si, si - 1, order, [si - (i + 1) for i in range(1, order)])
        if order == 1:
            rho0, rho_last = 0.0, 0.5
        else:
            rhos = torch.linalg.solve(c["R"], c["b"]).to(torch.float32)
            rho0, rho_last = rhos[0].item(), rhos[-1].item()
        out = torch.empty_like(last_sample)
        m_prev = self.model_outputs[-2] if order == 2 else None
        _lib.call("uv_unipc_corrector", _lib.ptr(last_sample), _lib.ptr(self.model_outputs[-1]), _lib.ptr(m_prev),
                  _lib.ptr(x0), _lib.ptr(out), c["r"], c["c1"], c["c2"], rho0, rho_last, c["rk"], order, out.numel(),
                  _lib.stream_ptr())
        return out

    def _predict(self, sample, order):
        si = self._step_index
        c = self._coeffs(si + 1, si, order, [si - i for i in range(1, order)])
        out = torch.empty_like(sample)
        m_prev = self.model_outputs[-2] if order == 2 else None
        _lib.call("uv_unipc_predictor", _lib.ptr(sample), _lib.ptr(self.model_outputs[-1]), _lib.ptr(m_prev), _lib.ptr(out),
                  c["r"], c["c1"], c["c2"], c["rk"], order, out.numel(), _lib.stream_ptr())
        return out

    def _advance(self, x0, timestep, sample):
        """Everything in step() after convert_model_output (:691-741)."""
        use_corrector = (self._step_index > 0 and self._step_index - 1 not in self.disable_corrector
                         and self.last_sample is not None)
        if use_corrector:
            sample = self._correct(x0, self.last_sample, self.this_order)
        for i in range(self.config.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = x0
        self.timestep_list[-1] = timestep
        if self.config.lower_order_final:
            this_order = min(self.config.solver_order, len(self._timesteps_host) - self._step_index)
        else:
            this_order = self.config.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev

    def _check(self, *tensors):
        for t in tensors:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.UnividHipError("UniPC: latents must be contiguous fp32 GPU tensors")

    def step(self, model_output, timestep, sample, return_dict: bool = True, generator=None):
        """fm_solvers_unipc.py:657-741."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        self._check(model_output, sample)
        if self._step_index is None:
            self._init_step_index(timestep)
        x0 = torch.empty_like(sample)
        sigma = self.sigmas[self._step_index].item()
        # gs = 0 makes the CFG stage the identity: x0 = sample - sigma * model_output (:323)
        _lib.call("uv_cfg_convert", _lib.ptr(model_output), _lib.ptr(model_output), _lib.ptr(sample), 0.0, sigma, None,
                  _lib.ptr(x0), x0.numel(), _lib.stream_ptr())
        prev = self._advance(x0, timestep, sample)
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
        prev = self._advance(x0, timestep, sample)
        return (prev, npred) if want_noise_pred else prev
