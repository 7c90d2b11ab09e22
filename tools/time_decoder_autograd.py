"""Forward and forward + backward of the 1024^2 / channel-multiplier-2 decoder with a graph wanted (features and latent require
grad, parameters frozen: the train_ae.py shape), E3DGE_DECODER_AUTOGRAD = packed (packed forward, library backward on recomputed
activations) vs library.   python tools/time_decoder_autograd.py  -> one JSON line (also gpurun_out/decoder_autograd.json)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

dev = "cuda:0"
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval()
g.requires_grad_(False)
dec = g.decoder
_, wd = syn.synthetic_inputs(1, seed=1, device=dev)
feats = 0.5 * torch.randn(1, 256, 64, 64, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
for backend in ("packed", "library"):
    os.environ["E3DGE_DECODER_AUTOGRAD"] = backend

    def fwd():
        f = feats.detach().requires_grad_(True)
        l = wd.detach().requires_grad_(True)
        img, _ = dec(f, [l], input_is_latent=True, randomize_noise=False)
        return img, f, l

    def fwd_bwd():
        img, f, l = fwd()
        img.square().mean().backward()
    res[backend] = {"forward_ms": round(timed(fwd), 4), "forward_backward_ms": round(timed(fwd_bwd), 4)}
with torch.no_grad():
    res["no_graph_forward_ms"] = round(timed(lambda: dec(feats, [wd], input_is_latent=True, randomize_noise=False)), 4)
line = json.dumps({"what": "decoder 64^2 -> 1024^2 under autograd (features + latent require grad, parameters frozen)", **res})
print(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/decoder_autograd.json", "w").write(line + "\n")
