"""Time the renderer's training direction (forward with saved arguments + HIP backward to the styles) at 64x64x24.
Usage: python tools/time_training.py [batch] [iters]     (wrap in rocprofv3 --kernel-trace --stats for per-kernel times)"""
import sys
import time

import torch

sys.path.insert(0, ".")
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import saved_state_buffer, VolumeFeatureRenderer  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
res, S = 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=res, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
r.requires_grad_(False)
wr, _ = syn.synthetic_inputs(batch, seed=7, device=dev)
poses, focal, near, far, _ = generate_camera_params(res, dev, batch=batch)
G = torch.randn(batch, 256, res, res, device=dev)
G_rgb = torch.randn(batch, 3, res, res, device=dev)


def step():
    s = wr.clone().requires_grad_(True)
    out = r(poses, focal, near, far, styles=s)
    ((out['features'] * G).sum() + (out['gen_thumb_imgs'] * G_rgb).sum()).backward()
    return s.grad


def ev_time(fn, n=iters, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fwd_infer():
    with torch.no_grad():
        r(poses, focal, near, far, styles=wr)


s_keep = wr.clone().requires_grad_(True)
out_keep = r(poses, focal, near, far, styles=s_keep)
loss_keep = (out_keep['features'] * G).sum() + (out_keep['gen_thumb_imgs'] * G_rgb).sum()


def fwd_save():
    r(poses, focal, near, far, styles=s_keep)


def bwd_only():
    s_keep.grad = None
    loss_keep.backward(retain_graph=True)


t_inf, t_save, t_bwd, t_step = ev_time(fwd_infer), ev_time(fwd_save), ev_time(bwd_only), ev_time(step)
rays = batch * res * res
print(f"batch {batch} (64x64x24): forward inference {t_inf:.3f} ms | forward saving arguments {t_save:.3f} ms | "
      f"backward (loss + composite_bwd + siren_bwd + reduce + film_bwd) {t_bwd:.3f} ms | whole step {t_step:.3f} ms "
      f"= {rays / t_step * 1e3:.3e} rays/s")

# ---- the backward launches alone (C-ABI level, preallocated buffers) ----
import ctypes  # noqa: E402
from e3dge_amd import _lib  # noqa: E402
from e3dge_amd.volume_renderer import siren_backward  # noqa: E402

film = r.siren.film_params(wr)
n_pts = res * res * S
args = saved_state_buffer(batch, n_pts, 9, dev)
with torch.no_grad():
    out = r.render_with_film(film, focal, poses, near, far, None, save_args=args)
d_rgb_pts = torch.randn(batch, n_pts, 3, device=dev)
d_sdf_pts = torch.randn(batch, n_pts, device=dev)
t_chain = ev_time(lambda: siren_backward(r.siren, film, args, None, d_rgb_pts, d_sdf_pts))
print(f"  e3dge_siren_bwd alone (MLP chain + reduce + film_bwd + zeroing partials): {t_chain:.3f} ms "
      f"-> {batch * n_pts * 8 * 2 * 256 * 256 / t_chain / 1e9:.1f} TFLOP/s on the 8 transposed GEMMs")
