"""Forward + backward of the 1024^2 / channel-multiplier-2 decoder in the shape train_ae.py's stage-1 step takes (trainer.py:1017-1031:
the feature map requires grad, the decoder latent and the generator's parameters do not), E3DGE_DECODER_AUTOGRAD = auto (packed forward +
e3dge_dec2_backward) vs library (weight modulation + MIOpen for both directions), plus the HIP-event time of every launch of the packed
backward.   python tools/time_decoder_autograd.py [--batch B]  -> one JSON line (also gpurun_out/decoder_autograd.json)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--latent-grad", action="store_true", help="also time the shape with the latent trainable too (native since round 5: per-channel sums)")
ap.add_argument("--only-latent", action="store_true", help="profiling: run only the packed forward + backward with d latent, 20 times")
args = ap.parse_args()
dev = "cuda:0"
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval()
g.requires_grad_(False)
dec = g.decoder
B = args.batch
_, wd = syn.synthetic_inputs(B, seed=1, device=dev)
feats = 0.5 * torch.randn(B, 256, 64, 64, device=dev)
gy = torch.randn(B, 3, 1024, 1024, device=dev) / 1024


def timed(fn, n=20):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:       # bring the clock up (it drops while the host is busy: DESIGN.md 5)
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
if args.only_latent:
    def fwd_bwd_l0():
        f = feats.detach().requires_grad_(True)
        l = wd.detach().requires_grad_(True)
        img, _ = dec(f, [l], input_is_latent=True, randomize_noise=False)
        torch.autograd.grad(img, [f, l], gy)
    print(json.dumps({"forward_backward_with_d_latent_ms": round(timed(fwd_bwd_l0), 4)}))
    sys.exit(0)
for backend in ("auto", "library"):
    os.environ["E3DGE_DECODER_AUTOGRAD"] = backend

    def fwd():
        f = feats.detach().requires_grad_(True)
        img, _ = dec(f, [wd], input_is_latent=True, randomize_noise=False)
        return img, f

    def fwd_bwd():
        img, f = fwd()
        torch.autograd.grad(img, [f], gy)
    res[backend] = {"forward_ms": round(timed(fwd), 4), "forward_backward_ms": round(timed(fwd_bwd), 4)}
    if args.latent_grad:
        def fwd_bwd_l():
            f = feats.detach().requires_grad_(True)
            l = wd.detach().requires_grad_(True)
            img, _ = dec(f, [l], input_is_latent=True, randomize_noise=False)
            torch.autograd.grad(img, [f, l], gy)
        res[backend]["forward_backward_with_d_latent_ms"] = round(timed(fwd_bwd_l), 4)
os.environ.pop("E3DGE_DECODER_AUTOGRAD", None)
with torch.no_grad():
    res["no_graph_forward_ms"] = round(timed(lambda: dec(feats, [wd], input_is_latent=True, randomize_noise=False)), 4)
    # per-launch times of one packed forward (save mode) + backward
    noise = [getattr(dec.noises, f"noise_{i}") for i in range(dec.num_layers)]
    kf, kb = [], []
    for _ in range(3):
        dec._forward_packed(feats, wd, noise, kernel_ms=kf, save=True)
        dec._backward_packed(feats, gy, kernel_ms=kb)
    res["forward_launches_ms"] = {n: round(v, 4) for n, v in zip(dec.dec2_launch_names(), kf)}
    res["backward_launches_ms"] = {n: round(v, 4) for n, v in zip(dec.dec2_bwd_launch_names(), kb)}
    res["backward_sum_of_launches_ms"] = round(sum(kb), 4)
    res["forward_sum_of_launches_ms"] = round(sum(kf), 4)
line = json.dumps({"what": f"decoder 64^2 -> 1024^2 under autograd, batch {B} (features require grad, latent and parameters frozen)", **res})
print(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/decoder_autograd.json", "w").write(line + "\n")
