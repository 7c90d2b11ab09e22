"""The 1024^2 / channel-multiplier-2 decoder forward (activations kept) + backward to the feature map under autograd, N times (stage-1
shape: features require grad, latent and parameters frozen): run under `rocprofv3 --kernel-trace --stats` / `--pmc` for the per-kernel
composition of one forward + backward.   python tools/decoder_bwd_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval()
g.requires_grad_(False)
_, wd = syn.synthetic_inputs(1, seed=1, device=dev)
feats = 0.5 * torch.randn(1, 256, 64, 64, device=dev)
gy = torch.randn(1, 3, 1024, 1024, device=dev) / 1024


def step():
    f = feats.detach().requires_grad_(True)
    img, _ = g.decoder(f, [wd], input_is_latent=True, randomize_noise=False)
    return torch.autograd.grad(img, [f], gy)[0]


for _ in range(5):
    step()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(iters):
    d = step()
b.record()
torch.cuda.synchronize()
print(f"decoder 64^2 -> 1024^2 forward + backward to the features: {a.elapsed_time(b) / iters:.3f} ms; d features {tuple(d.shape)}")
