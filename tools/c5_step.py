"""The stage-1 (C5) renderer step of bench.py alone -- 64x64 rays x 18 samples, eikonal + surface-normal terms, backward to the
styles -- N times: run under rocprofv3 (--kernel-trace --stats / --pmc FETCH_SIZE / WRITE_SIZE) for its per-kernel composition and
HBM traffic.   python tools/c5_step.py [iters] [samples per step]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import VolumeFeatureRenderer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda:0"
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=18), out_im_res=64, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
r.requires_grad_(False)
w, _ = syn.synthetic_inputs(B, seed=1, device=dev)
p, f, n, fa, _ = generate_camera_params(64, dev, locations=torch.zeros(B, 2, device=dev))


def step():
    s = w.clone().requires_grad_(True)
    o = r(p, f, n, fa, styles=s, return_eikonal=True, return_surface_eikonal=True)
    ((o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean() + (o['surface_eikonal_term'] ** 2).mean()).backward()
    return s.grad


import time  # noqa: E402
step()                                       # (the first step of a process pays one-off costs -- weight images, first launches -- and may alone exceed the window below)
torch.cuda.synchronize()
t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.4:      # the GPU clocks down while a process starts (imports: ~2 s of idling) and needs ~0.2 s of work to come
    step()                                   # back: three warm-up steps read 3.4-4.1 ms in every process but the first on a fresh box
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(iters):
    g = step()
b.record()
torch.cuda.synchronize()
print(f"stage-1 step {B} x 64x64x18: {a.elapsed_time(b) / iters:.3f} ms ({a.elapsed_time(b) / iters / B:.3f} per sample); |dstyles| max {float(g.abs().max()):.3e}")
