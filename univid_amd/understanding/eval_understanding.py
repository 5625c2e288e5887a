"""The frame ranker of UniVid's Pyramid-Reflection understanding loop: `Siglip2Scorer` and `mmr_select`
(reference models/BAGEL/eval_understanding.py:171-206 and :225-240), same names, signatures and return values.

Image pre-processing (NaFlex patchification, normalisation) and tokenisation are the HF processor's job in the reference
(`AutoProcessor.from_pretrained(ckpt)`); here the processor is injected (any callable with the HF processor's call
convention), so this module has no dependency on `transformers`. The towers run on the HIP kernels (siglip2.py); cosine
scoring = `uv_l2_normalize_rows_f32` + one small fp32 row product.
"""
from typing import List, Tuple

import torch

from .. import _lib
from .siglip2 import Siglip2Model


def _normalize(x: torch.Tensor) -> torch.Tensor:
    x = x.float().contiguous()
    out = torch.empty_like(x)
    _lib.call("uv_l2_normalize_rows_f32", _lib.ptr(x), x.stride(0), _lib.ptr(out), out.stride(0), x.shape[0], x.shape[1], 1e-12,
              _lib.stream_ptr())
    return out


class Siglip2Scorer:
    def __init__(self, ckpt: str = None, device: str = "cuda:0", dtype: torch.dtype = torch.float16, *, model: Siglip2Model = None,
                 processor=None):
        if dtype not in (torch.float16, torch.bfloat16):
            raise NotImplementedError("the SigLIP2 towers run with fp16 (reference default) or bf16 operands, fp32 accumulation")
        self.device = torch.device(device)
        self.dtype = dtype
        if processor is None:
            try:
                from transformers import AutoProcessor
            except ImportError as e:
                raise ValueError("pass processor= (images/text -> HF-style tensors) when transformers is not installed") from e
            processor = AutoProcessor.from_pretrained(ckpt)
        self.proc = processor
        if model is None:
            if ckpt is None:
                raise ValueError("pass model= (a Siglip2Model) or ckpt=")
            model = Siglip2Model.from_pretrained(ckpt)
        self.model = model.to(self.device).eval().set_operand_dtype(dtype)
        self.overlap_towers = True      # rank_frames: text tower on a side stream beside the vision tower (False = one after the other)
        self._side = None

    @torch.no_grad()
    def emb_text(self, q: str) -> torch.Tensor:
        t_inputs = self.proc(text=[q], return_tensors="pt")
        t = self.model.get_text_features(**{k: v for k, v in dict(t_inputs).items() if k in ("input_ids", "attention_mask")})
        return _normalize(t)

    @torch.no_grad()
    def emb_imgs(self, images: List, bs: int = 64) -> torch.Tensor:
        vecs = []
        for i in range(0, len(images), bs):
            x = dict(self.proc(images=images[i:i + bs], return_tensors="pt"))
            v = self.model.get_image_features(pixel_values=x["pixel_values"], pixel_attention_mask=x["pixel_attention_mask"],
                                              spatial_shapes=x["spatial_shapes"])
            vecs.append(_normalize(v))
        return torch.cat(vecs, dim=0) if vecs else torch.empty(0, self.model.config["vision"]["hidden_size"], device=self.device)

    @torch.no_grad()
    def rank_frames(self, frames: List, query: str, topk: int, bs: int = 64) -> Tuple[List[int], List[float]]:
        if len(frames) == 0:
            return [], []
        # The two towers are independent until the similarity: the text query (64 tokens: ~150 tiny, latency-bound launches, 1.7 ms if run by
        # itself in front of the frames) goes to a side stream and runs BESIDE the vision tower's large launches (round 6: rank_frames 6.6 ->
        # ~5 ms for 64 frames). Same kernels, same arithmetic per tower; every scratch tensor of a tower is its own allocation on its own stream.
        if self.overlap_towers and self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            self._side.wait_stream(cur)
            v = self.emb_imgs(frames, bs=bs)          # issued first: its launches keep the GPU busy while the host issues the query's
            with torch.cuda.stream(self._side):
                t = self.emb_text(query)
            cur.wait_stream(self._side)
            t.record_stream(cur)
        else:
            t = self.emb_text(query)
            v = self.emb_imgs(frames, bs=bs)
        sims = torch.empty(1, v.shape[0], dtype=torch.float32, device=v.device)
        # sims[0, i] = <t, v_i>: the fp32 row product kernel with the image embeddings as the "weight" rows
        _lib.call("uv_linear_rows_f32", _lib.ptr(t), t.stride(0), _lib.ptr(v), None, _lib.ptr(sims), sims.stride(0), 1, v.shape[0],
                  v.shape[1], 0, _lib.stream_ptr())
        sims = sims[0]
        k = min(topk, sims.shape[0])
        vals, idx = torch.topk(sims, k=k)
        return idx.tolist(), [float(x) for x in vals.tolist()]


def mmr_select(embs: torch.Tensor, query_emb: torch.Tensor, K: int, lam: float = 0.5) -> List[int]:
    """Greedy maximal-marginal-relevance selection (eval_understanding.py:225-240): the two similarity tables are one product
    each; the greedy loop is host logic over at most a few dozen frames (first maximum wins, candidates in ascending order)."""
    sims_q = (embs @ query_emb.T).squeeze(-1).float().cpu()
    sims_ii = (embs @ embs.T).float().cpu()
    N = embs.shape[0]
    selected: List[int] = []
    candidate = list(range(N))
    while len(selected) < min(K, N) and candidate:
        best_i, best_score = None, -1e9
        for i in candidate:
            div = 0.0 if not selected else float(sims_ii[i, selected].max())
            score = lam * float(sims_q[i]) - (1.0 - lam) * div
            if score > best_score:
                best_score, best_i = score, i
        selected.append(best_i)
        candidate.remove(best_i)
    return selected
