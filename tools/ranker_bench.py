"""BASELINE.json config 5: SigLIP2-base patch16 encode of 64 keyframes (Pyramid-Reflection ranker), frames/s on one GPU.

Random-init weights of the siglip2-base geometry (768-d, 12 layers, 12 heads x 64, 256 patches of 16x16x3), NaFlex inputs
pixel_values [64, 256, 768], full masks, 16x16 grids; one text query of 64 tokens. FLOPs per frame: 12 layers x (8 h^2 + 4 h f)
per token x 256 tokens + attention 4 L^2 h + patch embedding + pooling head = ~46 GFLOP.
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd.understanding import Siglip2Model, Siglip2Scorer

_lib.init()
dev = "cuda"
V = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, patch_size=16,
         num_patches=256, layer_norm_eps=1e-6)
T = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=32000,
         max_position_embeddings=64, projection_size=768, layer_norm_eps=1e-6)
with torch.device(dev):
    m = Siglip2Model(dict(vision=V, text=T))
m.init_weights(0).eval()
B, N = int(os.environ.get("FRAMES", 64)), 256
g = torch.Generator(device=dev).manual_seed(0)
pv = torch.randn(B, N, 768, device=dev, generator=g)
mask = torch.ones(B, N, dtype=torch.int64, device=dev)
shapes = torch.tensor([[16, 16]] * B, device=dev)
ids = torch.randint(0, 32000, (1, 64), device=dev, generator=g)


class Proc:
    def __call__(self, images=None, text=None, return_tensors="pt"):
        if text is not None:
            return {"input_ids": ids}
        i = torch.tensor(images)
        return {"pixel_values": pv[i], "pixel_attention_mask": mask[i], "spatial_shapes": shapes[i]}


sc = Siglip2Scorer(device=dev, model=m, processor=Proc())
h, f, L, nl = 768, 3072, 256, 12
flops_frame = nl * (L * (8 * h * h + 4 * h * f) + 4 * L * L * h) + 2 * L * 768 * h + (2 * L * 2 * h * h + 4 * L * h + 2 * h * h + 4 * h * f)
ref = None
for mode in ("eager", "hipGraph replay"):
    m.vision_model.use_graph = mode != "eager"
    for name, fn in (("get_image_features (64 frames)", lambda: m.get_image_features(pv, mask, shapes)),
                     ("rank_frames (text + 64 frames + top-8)", lambda: sc.rank_frames(list(range(B)), "q", 8))):
        out = fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        print(f"{mode:16s} {name:42s} {t * 1e3:8.2f} ms  {B / t:9.1f} frames/s  {B * flops_frame / t / 1e12:7.1f} TFLOP/s", flush=True)
        if torch.is_tensor(out):
            if ref is None:
                ref = out.clone()
            else:
                print("    graph replay bit-identical to eager:", bool(torch.equal(ref, out)))
