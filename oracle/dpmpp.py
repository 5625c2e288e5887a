"""TEST INFRASTRUCTURE ONLY (the checker; `univid_amd` must not import this).

CPU restatement of the flow-matching DPM-Solver++ sampler as WanTI2V drives it when `sample_solver='dpm++'`
(/root/reference/models/wan/textimage2video.py:343-351, 535-543; /root/reference/models/wan/utils/fm_solvers.py:
get_sampling_sigmas :24-28, retrieve_timesteps :31-68, ctor :131-201, set_timesteps :228-291, convert_model_output
:343-414, dpm_solver_first_order_update :417-485, multistep_dpm_solver_second_order_update :488-595, step :708-800),
specialised to the configuration the ctor defaults give: solver_order 2, dpmsolver++, midpoint, flow_prediction,
lower_order_final, final sigma 0, no thresholding.

As in the UniPC restatement every scalar coefficient is a 0-dim fp32 CPU tensor, so each `coef * tensor` is one fp32
rounding; pinned bit-exact against the imported reference by oracle/gen_golden.py (tests/golden/dpmpp.npz).
"""
import numpy as np
import torch


def get_sampling_sigmas(sampling_steps, shift):
    """fm_solvers.py:24-28 - float64 numpy, starts at sigma 1 (timestep 1000), already shifted."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


class FlowDPMpp:
    def __init__(self, num_train_timesteps=1000, shift=1.0, solver_order=2):
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.shift0 = shift
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                      # :184-187
        self.sigmas = sigmas
        self.sigma_min = sigmas[-1].item()
        self.sigma_max = sigmas[0].item()
        self.timesteps = sigmas * num_train_timesteps
        self.step_index = None

    def set_timesteps(self, num_inference_steps=None, sigmas=None, shift=None):
        """:228-291 - WanTI2V passes sigmas=get_sampling_sigmas(steps, shift) through retrieve_timesteps, so the ctor's
        shift (1) is applied on top as the identity."""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift0
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.step_index = None
        return self.timesteps

    def _lambda(self, sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _first(self, m0, sample):
        """:417-485 (dpmsolver++ branch)."""
        sigma_t, sigma_s = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s)
        return (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * m0

    def _second(self, sample):
        """:488-595 (dpmsolver++, midpoint)."""
        si = self.step_index
        sigma_t, sigma_s0, sigma_s1 = self.sigmas[si + 1], self.sigmas[si], self.sigmas[si - 1]
        alpha_t = 1 - sigma_t
        lambda_t, lambda_s0, lambda_s1 = self._lambda(sigma_t), self._lambda(sigma_s0), self._lambda(sigma_s1)
        m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
        h, h_0 = lambda_t - lambda_s0, lambda_s0 - lambda_s1
        r0 = h_0 / h
        d0, d1 = m0, (1.0 / r0) * (m0 - m1)
        return ((sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * d0
                - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * d1)

    def step(self, model_output, timestep, sample):
        """:708-800; returns prev_sample."""
        if self.step_index is None:
            idx = (self.timesteps == timestep).nonzero()
            self.step_index = idx[1 if len(idx) > 1 else 0].item()                  # :681-693
        n = len(self.timesteps)
        lower_order_final = self.step_index == n - 1                                # final_sigmas_type == "zero" (:771-774)
        lower_order_second = self.step_index == n - 2 and n < 15                    # :775-777
        x0 = sample - self.sigmas[self.step_index] * model_output                 # convert_model_output :393-394
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        sample = sample.to(torch.float32)
        if self.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev = self._first(x0, sample)
        else:                                                                       # solver_order == 2 (lower_order_second is moot)
            del lower_order_second
            prev = self._second(sample)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(x0.dtype)
