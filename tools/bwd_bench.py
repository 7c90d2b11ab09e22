"""Launch the backward chain (e3dge_siren_bwd) N times on a saved 64x64x24 forward; no autograd involved, so it can
run under rocprofv3 --kernel-trace --stats for per-kernel durations.   python tools/bwd_bench.py [batch] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import saved_state_buffer, VolumeFeatureRenderer, siren_backward  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev, res, S = "cuda:0", 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=res, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
r.requires_grad_(False)
wr, _ = syn.synthetic_inputs(batch, seed=7, device=dev)
poses, focal, near, far, _ = generate_camera_params(res, dev, batch=batch)
film = r.siren.film_params(wr)
n_pts = res * res * S
args = saved_state_buffer(batch, n_pts, 9, dev)
with torch.no_grad():
    r.render_with_film(film, focal, poses, near, far, None, save_args=args)
d_rgb = torch.randn(batch, n_pts, 3, device=dev)
d_sdf = torch.randn(batch, n_pts, device=dev)
d_feat = torch.randn(batch, n_pts, 256, device=dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf)
torch.cuda.synchronize()
a.record()
for _ in range(iters):
    siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf)
b.record()
torch.cuda.synchronize()
print(f"batch {batch}: e3dge_siren_bwd {a.elapsed_time(b) / iters:.3f} ms per call")
