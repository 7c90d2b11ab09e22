"""Run-to-run determinism of the inversion forward (pass #1, head + FiLM, pass #2, decoder), stage by stage: N eager runs and N
replays of the captured HIP graph, every stage's output compared BIT FOR BIT with the first eager run.  None of the kernels on
this path has a floating-point atomic, so any mismatch is a race.    python tools/determinism_check.py [N]"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
dec = gl.decoder


def stages():
    o1 = gl([w1, d1], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
    rec = vr._BACKBONE.get(gl.renderer)
    out = {"pass1.features": o1['features'].clone(), "pass1.weights": o1['hit_prob'].clone(), "record": rec['buf'].clone()}
    o2 = gl([w1, d1], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})
    rec = vr._BACKBONE.get(gl.renderer)
    if rec is not None and rec.get('tex_buf') is not None:
        out["film_record"] = rec['tex_buf'].clone()
    out["pass2.features"] = o2['features'].clone()
    for i in range(2 * len(dec.to_rgbs) + 1):          # packed activations of the decoder (the last one is not stored: fused ToRGB)
        try:
            out[f"decoder.act{i}"] = dec.dec2_unpack(i, (1, 256, RES, RES)).clone()
        except Exception:
            pass
    out["image"] = o2['gen_imgs'].clone()
    return out


with torch.no_grad():
    ref = stages()
    bad = {k: 0 for k in ref}
    worst = {k: 0.0 for k in ref}
    for _ in range(N):
        cur = stages()
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad[k] += 1
                if v.is_floating_point():
                    worst[k] = max(worst[k], float((v - ref[k]).abs().max()))
    def img(a, b):
        gl([a, b], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
        return gl([a, b], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})['gen_imgs']
    gi = GraphedCall(img, w1, d1)
    gbad, gworst = 0, 0.0
    for _ in range(N):
        x = gi(w1, d1)
        torch.cuda.synchronize()
        if not torch.equal(x, ref["image"]):
            gbad += 1
            gworst = max(gworst, float((x - ref["image"]).abs().max()))
line = json.dumps({"what": "determinism", "runs": N, "eager_mismatches": bad, "eager_worst_abs": worst, "graph_mismatches": gbad, "graph_worst_abs": gworst})
print(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/determinism.json", "a").write(line + "\n")
