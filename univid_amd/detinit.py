"""Deterministic, device-independent synthetic weights.

There are no checkpoints in this environment (no network), and torch's RNG streams differ between CPU
and GPU, so every test / bench / fixture that needs "random-init weights of the reference architecture"
derives them from a counter-based generator instead: value(name, index) = splitmix64(fnv1a(name) ^ seed
+ index) -> 24-bit uniform in [-1, 1) -> scaled per tensor role. Pure int64 tensor arithmetic, so the same
bits come out on CPU and on the MI355X, and a fixture only has to store the seed.

Scales follow WanModel.init_weights (models/wan/utils/modules/model.py:524-546) in spirit: Xavier-uniform
bound for Linear / patch-embedding weights, N(0, 0.02)-equivalent uniform for the text/time embedding
Linears, modulation ~ 1/sqrt(dim) (model.py:217,277). Unlike the reference's init, biases and the head
weight are NOT zero (zero weights would hide bias / head bugs from the parity tests).
"""
import math

import torch

_M64 = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _to_i64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, s):  # logical shift right on int64 tensors
    return (x >> s) & ((1 << (64 - s)) - 1)


def uniform_pm1(name: str, numel: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """numel float32 values in [-1, 1), a pure function of (name, seed, index)."""
    base = _to_i64(_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15))
    out = torch.empty(numel, dtype=torch.float32, device=device)
    step = 1 << 24
    c1, c2, g = _to_i64(0xBF58476D1CE4E5B9), _to_i64(0x94D049BB133111EB), _to_i64(0x9E3779B97F4A7C15)
    for s in range(0, numel, step):
        n = min(step, numel - s)
        z = (torch.arange(s + 1, s + n + 1, dtype=torch.int64, device=device) * g) + base
        z = (z ^ _lsr(z, 30)) * c1
        z = (z ^ _lsr(z, 27)) * c2
        z = z ^ _lsr(z, 31)
        out[s:s + n] = (_lsr(z, 40).to(torch.float32) * (2.0 ** -23)) - 1.0
    return out


def _role_scale(name: str, shape) -> float:
    """Half-width of the uniform distribution for a parameter, by its role."""
    if name.endswith("modulation"):
        return math.sqrt(3.0) / math.sqrt(shape[-1])
    if name.endswith(".gamma"):
        return 0.0  # handled as 1 + small
    if name.endswith(".bias"):
        return 0.02
    if "norm" in name and name.endswith(".weight"):
        return 0.0  # handled as 1 + small
    if name.startswith(("text_embedding", "time_embedding")):
        return 0.02 * math.sqrt(3.0)
    if name.startswith("head.head"):
        return 0.02 * math.sqrt(3.0)
    if len(shape) >= 2:
        fan_out = shape[0]
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        if len(shape) > 2:  # conv: receptive field multiplies both fans
            rf = 1
            for d in shape[2:]:
                rf *= d
            fan_out = shape[0] * rf
        return math.sqrt(6.0 / (fan_in + fan_out))
    return 0.02


def fill_(name: str, tensor: torch.Tensor, seed: int = 0) -> torch.Tensor:
    """In-place deterministic init of one parameter (float32 values, cast to the tensor's dtype)."""
    u = uniform_pm1(name, tensor.numel(), seed, device=tensor.device).view(tensor.shape)
    if (("norm" in name and name.endswith(".weight")) or name.endswith(".gamma")):
        v = 1.0 + 0.1 * u
    else:
        v = u * _role_scale(name, tuple(tensor.shape))
    with torch.no_grad():
        tensor.copy_(v.to(tensor.dtype))
    return tensor


def init_state_dict_(sd: dict, seed: int = 0) -> dict:
    """Deterministically (re)initialises every floating tensor of a state dict, by key name."""
    for k, v in sd.items():
        if torch.is_floating_point(v):
            fill_(k, v, seed)
    return sd


def init_module_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    init_state_dict_(module.state_dict(keep_vars=True), seed)
    return module
