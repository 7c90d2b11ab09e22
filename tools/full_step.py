"""The step train_ae.py runs for one stage-1 sample -- G_pred_latents.forward (renderer 64x64x18 with both eikonal terms -> decoder 64^2 ->
1024^2), pixel loss on pool_256(gen_imgs) + the renderer losses, loss.backward() to the styles -- N times: run under rocprofv3
(--kernel-trace) and lay out with tools/step_timeline.py for the step's timeline.   python tools/full_step.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=18), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval()
g.requires_grad_(False)
w, d = syn.synthetic_inputs(1, seed=1, device=dev)
p, f, n, fa, _ = generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
pool = torch.nn.AdaptiveAvgPool2d((256, 256))


def step():
    s = w.clone().requires_grad_(True)
    o = g([s, d], p, f, n, fa, input_is_latent=True, randomize_noise=False, return_eikonal=True, return_surface_eikonal=True)
    ((pool(o['gen_imgs']) ** 2).mean() + (o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean()
     + (o['surface_eikonal_term'] ** 2).mean()).backward()
    return s.grad


for _ in range(3):
    step()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(iters):
    gr = step()
b.record()
torch.cuda.synchronize()
print(f"full stage-1 step 64x64x18 + decoder 1024^2: {a.elapsed_time(b) / iters:.3f} ms; |dstyles| max {float(gr.abs().max()):.3e}")
