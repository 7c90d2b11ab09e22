"""C3 evaluation loop timed repeatedly with one and two issue streams (tools; see bench.py's c3 leg for the driver's figure).
    python tools/time_c3.py [n_images]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import sharded_eval, synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24, enable_local_model=True, L_pred_tex_modulations=True),
                    full_pipeline=True)
syn.load_synthetic(gl)
gl = gl.to(dev).eval()
gl.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
feats = syn.synthetic_local_feats(1, 64, 24, device=dev)
target = torch.tanh(torch.randn(1, 3, 1024, 1024, device=dev))
codes = [syn.synthetic_inputs(1, seed=1000 + i, device=dev) for i in range(n)]


def unit(i):
    w_r, w_d = codes[i]
    gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
    o = gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})
    return sharded_eval.image_metrics(o['gen_imgs'], target)


with torch.no_grad():
    for i in range(2):
        unit(i)
    for ns in (1, 2, 1, 2, 3, 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sharded_eval.evaluate_sharded(unit, n, 0, 1, device=dev, n_streams=ns)
        torch.cuda.synchronize()
        print(f"{ns} stream(s): {1e3 * (time.perf_counter() - t0) / n:.3f} ms per image")
