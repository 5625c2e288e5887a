"""TEST INFRASTRUCTURE ONLY (the checker; `univid_amd` must not import this).

CPU restatement of the flow-matching UniPC sampler exactly as UniVid drives it
(/root/reference/models/wan/utils/fm_solvers_unipc.py; ctor :79-134, set_timesteps :162-229,
convert_model_output :281-350, multistep_uni_p_bh_update :352-486, multistep_uni_c_bh_update :488-628,
step :657-741), specialised to the only configuration WanTI2V uses (textimage2video.py:336-342):
solver_order=2, bh2, predict_x0, flow_prediction, lower_order_final, final sigma 0, no thresholding.

`step()` runs inside the pipeline's ambient autocast(bf16) (textimage2video.py:329-333), but nothing in
it is autocast-eligible: the history term `einsum('k,bkc...->bc...', rhos, D1s)` contracts a size-1 axis,
which ATen evaluates as an elementwise multiply (no bmm), so the whole update stays fp32 (probed:
reference step() under autocast == without autocast, bit for bit).
All scalar coefficients are 0-dim fp32 CPU tensors, as in the reference (sigmas live on the CPU, :131,228);
every `coef * tensor` is therefore one fp32 rounding of (fp32 scalar x element).
"""
import numpy as np
import torch

BF16 = torch.bfloat16


def _einsum_ac(rhos, d1s):
    """einsum('k,bkc...->bc...') over the single history term (k = 1): an fp32 scale by rhos[0]."""
    return torch.einsum("k,bkc...->bc...", rhos, d1s)


class FlowUniPC:
    def __init__(self, num_train_timesteps=1000, shift=1.0, solver_order=2):
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.shift0 = shift
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                      # :116
        self.sigmas = sigmas
        self.sigma_min = sigmas[-1].item()
        self.sigma_max = sigmas[0].item()
        self.timesteps = sigmas * num_train_timesteps

    def set_timesteps(self, num_inference_steps, shift=None):
        """:162-229 - linspace in float64 numpy, shift, int64-truncated timesteps, trailing sigma 0."""
        sig = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift0
        sig = shift * sig / (1 + (shift - 1) * sig)
        timesteps = sig * self.num_train_timesteps
        sig = np.concatenate([sig, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = None
        self.this_order = None
        return self.timesteps

    # ---- coefficient algebra shared by predictor and corrector (:407-455 / :550-598)
    def _coeffs(self, i_t, i_s0, order, hist_offsets):
        sigma_t, sigma_s0 = self.sigmas[i_t], self.sigmas[i_s0]
        alpha_t, alpha_s0 = 1 - sigma_t, 1 - sigma_s0
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = []
        for si in hist_offsets:
            sigma_si = self.sigmas[si]
            lambda_si = torch.log(1 - sigma_si) - torch.log(sigma_si)
            rks.append((lambda_si - lambda_s0) / h)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        R, b = [], []
        rks_t = torch.tensor(rks + [1.0])
        factorial_i = 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        return dict(sigma_t=sigma_t, sigma_s0=sigma_s0, alpha_t=alpha_t, h_phi_1=h_phi_1, B_h=B_h, rks=rks,
                    R=torch.stack(R), b=torch.tensor(b))

    def _predict(self, sample, order):
        """multistep_uni_p_bh_update :352-486."""
        m0 = self.model_outputs[-1]
        si = self.step_index
        c = self._coeffs(si + 1, si, order, [si - i for i in range(1, order)])
        x_t_ = c["sigma_t"] / c["sigma_s0"] * sample - c["alpha_t"] * c["h_phi_1"] * m0
        if order == 2:
            d1s = torch.stack([(self.model_outputs[-2] - m0) / c["rks"][0]], dim=1)
            pred_res = _einsum_ac(torch.tensor([0.5], dtype=sample.dtype, device=sample.device), d1s)      # :460-461, 471
        else:
            pred_res = 0
        return (x_t_ - c["alpha_t"] * c["B_h"] * pred_res).to(sample.dtype)

    def _correct(self, model_t, last_sample, order):
        """multistep_uni_c_bh_update :488-628."""
        m0 = self.model_outputs[-1]
        si = self.step_index
        c = self._coeffs(si, si - 1, order, [si - (i + 1) for i in range(1, order)])
        if order == 1:
            rhos_c = torch.tensor([0.5], dtype=last_sample.dtype, device=last_sample.device)                   # :606-607
        else:
            rhos_c = torch.linalg.solve(c["R"], c["b"]).to(last_sample.dtype).to(last_sample.device)       # :609 (solved on the host)
        x_t_ = c["sigma_t"] / c["sigma_s0"] * last_sample - c["alpha_t"] * c["h_phi_1"] * m0
        if order == 2:
            d1s = torch.stack([(self.model_outputs[-2] - m0) / c["rks"][0]], dim=1)
            corr_res = _einsum_ac(rhos_c[:-1], d1s)
        else:
            corr_res = 0
        d1_t = model_t - m0
        return (x_t_ - c["alpha_t"] * c["B_h"] * (corr_res + rhos_c[-1] * d1_t)).to(last_sample.dtype)

    def step(self, model_output, timestep, sample):
        """step :657-741; returns prev_sample."""
        if self.step_index is None:
            idx = (self.timesteps == timestep).nonzero()
            self.step_index = idx[1 if len(idx) > 1 else 0].item()                  # :630-642
        use_corrector = self.step_index > 0 and self.last_sample is not None
        x0 = sample - self.sigmas[self.step_index] * model_output                 # convert_model_output :323
        if use_corrector:
            sample = self._correct(x0, self.last_sample, self.this_order)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.solver_order, len(self.timesteps) - self.step_index)  # lower_order_final :714-717
        self.this_order = min(this_order, self.lower_order_nums + 1)                 # warm-up :721
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev
