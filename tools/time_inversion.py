"""Stage timing of one inversion forward (pass #1 render, pass #2 render with texture FiLM, decoder 64^2 -> 1024^2):
device time by HIP events and host wall time, to tell GPU-bound from launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.stylesdf_model import G_pred_latents

dev = "cuda:0"
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24, enable_local_model=True), full_pipeline=True)
syn.load_synthetic(g)
g = g.to(dev).eval()
w, d = syn.synthetic_inputs(1, device=dev)
p, f, n, fa, _ = generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
tex = syn.synthetic_tex_conditions(1, 64, 24, device=dev)


def stage(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, 1e3 * t_host / reps, out


with torch.no_grad():
    t1 = stage(lambda: g([w, d], p, f, n, fa, input_is_latent=True, sample_with_renderer=True))
    t2 = stage(lambda: g([w, d], p, f, n, fa, input_is_latent=True, renderer_only=True, local_data_batch={'tex': tex}))
    feats = t2[2]['features']
    t3 = stage(lambda: g.decoder(feats, [d], input_is_latent=True, randomize_noise=False))
    conv = g.decoder.convs[6]     # 64 -> 32 @ 1024^2 upsampling StyledConv
    x = torch.randn(1, 64, 512, 512, device=dev)
    t4 = stage(lambda: conv(x, d[:, 7], noise=g.decoder.noises.noise_7))
    t5 = stage(lambda: conv.conv(x, d[:, 7]))
    import torch.nn.functional as F
    wt = conv.conv._weights(conv.conv.modulation(d[:, 7]), transpose=True)
    t6 = stage(lambda: F.conv_transpose2d(x, wt, padding=0, stride=2))
    conv2 = g.decoder.convs[7]
    y = torch.randn(1, 32, 1024, 1024, device=dev)
    w2 = conv2.conv._weights(conv2.conv.modulation(d[:, 8]), transpose=False)
    t7 = stage(lambda: F.conv2d(y, w2, padding=1))
for name, t in [("pass#1 render", t1), ("pass#2 render + tex FiLM", t2), ("decoder 64^2->1024^2", t3),
                ("  StyledConv up 64->32 @1024 (whole)", t4), ("    its ModulatedConv2d (convT + blur)", t5),
                ("      conv_transpose2d alone (MIOpen)", t6), ("  conv2d 3x3 32->32 @1024 alone (MIOpen)", t7)]:
    print(f"{name:<44} device {t[0]:7.3f} ms   host {t[1]:7.3f} ms")
