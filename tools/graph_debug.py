"""Debug: bench.py's sequence (eager inversion, the per-launch timing calls, then capture + replay) with stage outputs captured inside
the graph, to find which stage deviates under replay."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd import volume_renderer as vr  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.graphs import GraphedCall  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

dev = "cuda:0"
RES, S = 64, 24
g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S), full_pipeline=True)
syn.load_synthetic(g)
sd = {k: v.clone() for k, v in g.state_dict().items()}
gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), full_pipeline=True)
sd_l = {k.replace('renderer.network.', 'renderer.network.netGlobal.'): v for k, v in sd.items()}
for k, v in gl.state_dict().items():
    if '.netLocal.' in k:
        sd_l[k] = 0.05 * syn.synthetic_tensor(k, v.shape)
gl.load_state_dict(sd_l)
gl = gl.to(dev).eval()
gl.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
feats = syn.synthetic_local_feats(1, RES, S, device=dev)
w1, d1 = syn.synthetic_inputs(1, seed=1, device=dev)
rr = gl.renderer
head = rr.network.netLocal.local_feat_to_tex_modulations_linear
mode = sys.argv[1] if len(sys.argv) > 1 else "table"


def inversion(a, b):
    o1 = gl([a, b], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
    o2 = gl([a, b], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})
    return o1['features'], o1['hit_prob'], o2['features'], o2['gen_imgs']


with torch.no_grad():
    for _ in range(3):
        ref = [t.clone() for t in inversion(w1, d1)]
    if "heat" in mode:
        film_h = rr.siren.film_params(w1)
        for _ in range(3000):
            rr.render_with_film(film_h, f1, p1, n1, fa1)
        torch.cuda.synchronize()
    if "table" in mode:
        film = rr.siren.film_params(w1)
        tex = head.tex_modulations(feats)
        key = rr._reuse_key(w1, f1, p1, n1, fa1)
        rr.render_with_film(film, f1, p1, n1, fa1, reuse_key=key)
        rec = vr._BACKBONE.get(rr)
        tbuf = torch.zeros_like(rec['buf'])
        head.tex_film(feats, rec['buf'], tbuf, 1, RES, RES, S)
        rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=vr._LazyTex(head, feats), reuse_key=key)
        rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=tex, reuse_key=key)
        rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=tex)
        dec = gl.decoder
        latent, noise = dec.styles_and_noise_forward([d1], None, input_is_latent=True, randomize_noise=False)
        ms = []
        dec._forward_packed(ref[2].contiguous(), latent, noise, kernel_ms=ms)
    if "bench" in mode:      # exactly bench.py's function: the first call's outputs are dropped before the second call runs
        def inv_b(a, b):
            gl([a, b], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
            return gl([a, b], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})
        gi = GraphedCall(lambda a, b: inv_b(a, b)['gen_imgs'], w1, d1)
    elif "one" in mode:
        gi = GraphedCall(lambda a, b: inversion(a, b)[3], w1, d1)
    else:
        gi = GraphedCall(inversion, w1, d1)
    res = []
    for i in range(5):
        out = gi(w1, d1)
        torch.cuda.synchronize()
        if "one" in mode or "bench" in mode:
            res.append([float((out - ref[3]).abs().max())])
        else:
            res.append([float((a - b).abs().max()) for a, b in zip(out, ref)])
    eager_after = [float((a - b).abs().max()) for a, b in zip(inversion(w1, d1), ref)]
print(json.dumps({"mode": mode, "graph_vs_eager [pass1.features, weights, pass2.features, image] per replay": res, "eager_after": eager_after}))
