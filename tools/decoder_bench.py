"""The 1024^2 / channel-multiplier-2 decoder forward alone (fixed noise), N times: run under `rocprofv3 --kernel-trace --stats`
for the per-kernel composition of one decoder pass.   python tools/decoder_bench.py [iters]"""
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
with torch.no_grad():
    for _ in range(5):
        g.decoder(feats, [wd], input_is_latent=True, randomize_noise=False)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        img, _ = g.decoder(feats, [wd], input_is_latent=True, randomize_noise=False)
    b.record()
    torch.cuda.synchronize()
print(f"decoder 64^2 -> 1024^2: {a.elapsed_time(b) / iters:.3f} ms per pass; image {tuple(img.shape)}")
