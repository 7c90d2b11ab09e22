"""The stage-1 step through G_pred_latents at four samples per GPU (both latents trainable), per-iteration wall times: does it settle?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.stylesdf_model import G_pred_latents
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=18), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval(); g.requires_grad_(False)
w, d = syn.synthetic_inputs(B, seed=3, device=dev)
p, f, n, fa, _ = generate_camera_params(64, dev, locations=torch.zeros(B, 2, device=dev))
pool = torch.nn.AdaptiveAvgPool2d((256, 256))
def step():
    s_, d_ = w.clone().requires_grad_(True), d.clone().requires_grad_(True)
    o = g([s_, d_], p, f, n, fa, input_is_latent=True, randomize_noise=False, return_eikonal=True, return_surface_eikonal=True)
    loss = ((pool(o['gen_imgs']) ** 2).mean() + (o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean() + (o['surface_eikonal_term'] ** 2).mean())
    loss.backward()
ts = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
print("per-step ms:", ts, " reserved GB:", round(torch.cuda.memory_reserved() / 2**30, 2), " allocated peak GB:", round(torch.cuda.max_memory_allocated() / 2**30, 2))
