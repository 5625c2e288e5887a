"""TEST INFRASTRUCTURE ONLY (the checker; `univid_amd` must not import this).

CPU restatement of an UN-MERGED PEFT LoRA linear as the reference runs it at inference (LoRAManager.load_lora_weights ->
PeftModel.from_pretrained, /root/reference/models/model_pipeline.py:724-750; wrapped layers chosen at :470-560; the DiT forward
runs under autocast(bf16), models/wan/textimage2video.py:329-333).

THIRD-PARTY ARITHMETIC, PARITY UNPINNED: the algorithm lives in `peft` (pinned peft==0.17.1 in the reference's
environment.yaml:417), which is absent from /root/reference and from this image, so it can neither be imported nor executed here.
Restated from its published source (peft/tuners/lora/layer.py, `Linear.forward`, non-DoRA, eval mode so dropout is the identity):

    result = base_layer(x)                                    # nn.Linear under autocast -> bf16
    x = x.to(lora_A.weight.dtype)                             # fp32 adapter weights: no-op for the fp32 / bf16 activations here
    result = result + lora_B(lora_A(dropout(x))) * scaling    # both Linears under autocast -> bf16; bf16 * python float -> bf16
    result = result.to(torch_result_dtype)                    # bf16

with scaling = lora_alpha / r (lora_alpha / sqrt(r) with use_rslora), `LoraLayer.update_layer`.
"""
import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def lora_linear_ac(x, w, b, lora_a, lora_b, scaling):
    """One LoRA-wrapped nn.Linear under autocast(bf16), adapter un-merged."""
    base = F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))
    low = F.linear(F.linear(x.to(BF16), lora_a.to(BF16)), lora_b.to(BF16))
    return (base + low * scaling).to(base.dtype)


def merged_weight(w, lora_a, lora_b, scaling):
    """peft Linear.get_delta_weight / merge: what merge_and_unload() leaves in the dense layer (fp32)."""
    return w + (lora_b.float() @ lora_a.float()) * scaling
