"""Reads the cycle counts a -DE3DGE_BWD_TIMING build of the backward kernel leaves in dfilm (E3DGE_LIB_PATH must point
at that variant): per-wave averages over the workgroups, 64x64x24, batch 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import _lib, synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import saved_state_buffer, VolumeFeatureRenderer, siren_backward  # noqa: E402

dev, res, S = "cuda:0", 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=res, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
r.requires_grad_(False)
wr, _ = syn.synthetic_inputs(1, seed=7, device=dev)
poses, focal, near, far, _ = generate_camera_params(res, dev)
film = r.siren.film_params(wr)
n_pts = res * res * S
args = saved_state_buffer(1, n_pts, 9, dev)
with torch.no_grad():
    r.render_with_film(film, focal, poses, near, far, None, save_args=args)
d_rgb, d_sdf, d_feat = torch.randn(1, n_pts, 3, device=dev), torch.randn(1, n_pts, device=dev), torch.randn(1, n_pts, 256, device=dev)
for _ in range(3):
    _, dfilm = siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf)
torch.cuda.synchronize()
n_wg = _lib.load().e3dge_siren_bwd_partial_floats(1, n_pts) // (9 * 2 * 256)
v = (dfilm.reshape(-1)[:7] / n_wg).tolist()
names = ["total", "prologue(view layer)", "GEMM tiles (incl. sync+fetch)", "epilogues", "layer tails", "  of GEMM: vmcnt wait", "  of GEMM: barrier wait"]
print(f"{n_wg} workgroups; cycles per wave (s_memtime units), wave 0 average:")
for n, x in zip(names, v):
    print(f"  {n:<32} {x:12.0f}  ({100 * x / v[0]:.1f}%)")
print(f"  pure MFMA time would be {192 * 8192} shader cycles")
