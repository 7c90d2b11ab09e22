"""A/B of the backward-direction kernels: first generation (mode f16x3: 4 waves x 32 points) vs the 8-wave layout (f16x3_g2, round 6) vs fp32,
on a saved 64x64x24 forward, per entry point: time per call and the largest deviation from the fp32 kernels relative to the
largest fp32 value of the same output.   python tools/bwd_ab.py [batch] [iters]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import (saved_state_buffer, saved_state_point_major, VolumeFeatureRenderer, sdf_gradient, siren_backward,  # noqa: E402
                                       tangent_arguments)

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev, res, S = "cuda:0", 64, int(os.environ.get("BWD_AB_S", "24"))
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=res, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
r.requires_grad_(False)
wr, _ = syn.synthetic_inputs(batch, seed=7, device=dev)
poses, focal, near, far, _ = generate_camera_params(res, dev, batch=batch)
film = r.siren.film_params(wr)
n_pts = res * res * S
def saved_arguments():
    """The forward's saved pre-sine arguments in the layout the CURRENT backward mode reads (slab-major for f16x3_g2)."""
    a_ = saved_state_buffer(batch, n_pts, 9, dev)
    with torch.no_grad():
        r.render_with_film(film, focal, poses, near, far, None, save_args=a_)
    return a_
g = torch.Generator(device=dev).manual_seed(3)
d_rgb = torch.randn(batch, n_pts, 3, device=dev, generator=g)
d_sdf = torch.randn(batch, n_pts, device=dev, generator=g)
d_feat = torch.randn(batch, n_pts, 256, device=dev, generator=g)
v = torch.randn(batch, n_pts, 3, device=dev, generator=g)
tex_alpha = 0.05 * torch.randn(batch, n_pts, 256, device=dev, generator=g)
box = 1.0 / 0.12


def timed(fn):
    for _ in range(2):
        out = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters, out


def flat(out):
    if out is None:
        return []
    if torch.is_tensor(out):
        return [out]
    res_ = []
    for o in out:
        res_ += flat(o)
    return res_


results, ref = {}, {}
for mode in os.environ.get("BWD_AB_MODES", "f32,f16x3,f16x3_g2").split(","):
    r.siren.bwd_mode = mode
    r.siren.mfma_mode = "f32" if mode == "f32" else "f16x3"
    args = saved_arguments()
    eik, rsave = sdf_gradient(r.siren, film, args, box)
    tang, rs_ = tangent_arguments(r.siren, film, args, v, box, rsave=rsave)      # (f16x3_g2: tang = the products ta r, rs_ None)
    cases = {
        "bwd": lambda: siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf),
        "bwd_dpts": lambda: siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf, want_d_pts=True, box_scale=box),
        "bwd_eik": lambda: siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf, tang=tang, rsave=rs_),
        "bwd_eik_dpts": lambda: siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf, tang=tang, rsave=rs_, want_d_pts=True, box_scale=box),
        "bwd_tex": lambda: siren_backward(r.siren, film, args, d_feat, d_rgb, d_sdf, tex_alpha=tex_alpha),
        "sdf_grad": lambda: (lambda e_, r_: (e_, saved_state_point_major(r_, mode == "f16x3_g2")))(*sdf_gradient(r.siren, film, args, box)),
        "tangent": lambda: saved_state_point_major(tangent_arguments(r.siren, film, args, v, box)[0], mode == "f16x3_g2"),
        "tangent_tr": lambda: (saved_state_point_major(tangent_arguments(r.siren, film, args, v, box, rsave=rsave)[0], True) if mode == "f16x3_g2"
                               else tangent_arguments(r.siren, film, args, v, box)[0] * rsave),
        "save_fwd": lambda: saved_arguments()[:, :1, :1],
    }
    for name, fn in cases.items():
        ms, out = timed(fn)
        outs = [o.clone() for o in flat(out)]
        entry = {"ms": round(ms, 4)}
        if mode == "f32":
            ref[name] = outs
        elif name in ref:
            entry["rel_dev_vs_f32"] = [float(((o - q).abs().max() / q.abs().max().clamp_min(1e-30))) for o, q in zip(outs, ref[name])]
            entry["finite"] = all(bool(torch.isfinite(o).all()) for o in outs)
        results[f"{mode}/{name}"] = entry
        print(f"{mode:9s} {name:9s} {json.dumps(entry)}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open(os.environ.get("BWD_AB_OUT", "gpurun_out/bwd_ab.json"), "w") as f:
    json.dump({"batch": batch, "n_pts": n_pts, "results": results}, f, indent=1)
