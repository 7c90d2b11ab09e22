"""The stage-2 training step of bench.py alone (64x64x24 points, two (1,256,128,128) feature maps, Fuse_sft_MLP and the texture head
trainable, generator frozen), N times: for rocprofv3 --kernel-trace and event timing.   python tools/stage2_step.py [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.local_query import Fuse_sft_MLP
from e3dge_amd.stylesdf_model import G_pred_latents
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, RES, S = "cuda:0", 64, 24
g0 = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S), full_pipeline=True)
syn.load_synthetic(g0)
sd_cpu = g0.state_dict()
gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), full_pipeline=True)
sd_l = {k.replace('renderer.network.', 'renderer.network.netGlobal.'): v for k, v in sd_cpu.items()}
for k, v in gl.state_dict().items():
    if '.netLocal.' in k:
        sd_l[k] = 0.05 * syn.synthetic_tensor(k, v.shape)
gl.load_state_dict(sd_l)
gl = gl.to(dev).eval(); gl.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
w1, d1 = syn.synthetic_inputs(1, seed=1, device=dev)
fu2 = Fuse_sft_MLP().to(dev)
with torch.no_grad():
    for prm in fu2.parameters():
        prm.copy_(torch.randn_like(prm) * (0.1 if prm.ndim == 1 else 1.0 / prm.shape[1] ** 0.5))
fu2.requires_grad_(True)
head_ = gl.renderer.network.netLocal.local_feat_to_tex_modulations_linear
head_.requires_grad_(True)
gen = torch.Generator(device=dev).manual_seed(11)
maps2 = {k: torch.randn(1, 256, 128, 128, device=dev, generator=gen).requires_grad_(True) for k in ("ref", "que")}
cq_ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev), return_calibs=True)['calibs']
cr_ = generate_camera_params(RES, dev, locations=torch.tensor([[-0.2, 0.05]], device=dev), return_calibs=True)['calibs']
pool2 = torch.nn.AdaptiveAvgPool2d((256, 256))


def step():
    with torch.no_grad():
        o1 = gl.renderer(p1, f1, n1, fa1, styles=w1)
    s_ = w1.clone().requires_grad_(True)
    for t_ in list(maps2.values()) + list(fu2.parameters()) + list(head_.parameters()):
        t_.grad = None
    o2 = gl([s_, d1], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False,
            local_data_batch=dict(feature_maps=maps2, ref_calibs=cr_, que_calibs=cq_, points=o1['points'], xyz=o1['xyz'], fuse_sft_block=fu2))
    ((pool2(o2['gen_imgs']) ** 2).mean() + (o2['gen_thumb_imgs'] ** 2).mean()).backward()
    return s_.grad


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    g = step()
b.record(); torch.cuda.synchronize()
print(f"stage-2 step 64x64x24: {a.elapsed_time(b) / iters:.3f} ms; |dstyles| max {float(g.abs().max()):.3e}")
