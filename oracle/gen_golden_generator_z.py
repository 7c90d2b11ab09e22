"""Golden vectors for two generator entry points the first fixtures left out, recorded from the REAL reference (imported
via oracle/ref_harness.py) -- authoring container only.  TEST INFRASTRUCTURE.

    python oracle/gen_golden_generator_z.py       # writes tests/golden/generator_z_base.npz (+ generator_z_base_report.json)

(1) G_pred_latents.forward with a z-space input and truncation (stylesdf_model.py:1023-1172, styles_and_noise_forward
    :869-903; decoder side :692-740): input_is_latent=False, truncation=0.7, truncation_latent = [renderer mean, decoder mean]
    built from a seeded batch of z the way Generator.mean_latent does (:854-864).  size 256, cm 1, 64x64x24, fixed noise.
(2) The base Generator.forward (:923-1020) as the surface-extraction generator uses it (train_setup.py:112-126):
    full_pipeline=False, renderer 16x16 rays x 16 samples, return_sdf / return_xyz -> (None, thumb, xyz, sdf, mask)."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")

import e3dge_amd  # noqa: E402,F401
from e3dge_amd import synthetic as syn  # noqa: E402
from oracle import ref_harness  # noqa: E402
from oracle.gen_golden import build_reference_generator, npf, save  # noqa: E402


def main():
    torch.set_num_threads(8)
    vr, sm, cu, op = ref_harness.modules()
    report = {}
    # ------------------------------------------------------------------ (1) z input + truncation
    g, sd = build_reference_generator(sm, 256, 1, 24, 64)
    rs = np.random.RandomState(5)
    z = torch.from_numpy(rs.standard_normal((1, 256)).astype(np.float32))
    z_mean = torch.from_numpy(rs.standard_normal((64, 256)).astype(np.float32))
    c = cu.generate_camera_params(64, 'cpu', locations=torch.tensor([[0.2, -0.1]]), fov_ang=6, dist_radius=0.12)
    with torch.no_grad():
        mean_r = g.style(z_mean).mean(0, keepdim=True)                 # Generator.mean_latent :854-864 on a fixed batch
        mean_d = g.decoder.mean_latent(mean_r)
        out = g([z], c[0], c[1], c[2], c[3], input_is_latent=False, truncation=0.7, truncation_latent=[mean_r, mean_d],
                randomize_noise=False)
        plain = g([z], c[0], c[1], c[2], c[3], input_is_latent=False, randomize_noise=False)
    report['z_truncation'] = dict(styles_shape=list(out['styles'].shape), gen_imgs_abs_max=float(out['gen_imgs'].abs().max()),
                                  truncation_effect_on_image=float((out['gen_imgs'] - plain['gen_imgs']).abs().max()))
    arrays = dict(z=npf(z), z_mean=npf(z_mean), poses=npf(c[0]), focal=npf(c[1]), near=npf(c[2]), far=npf(c[3]),
                  ref_mean_r=npf(mean_r), ref_mean_d=npf(mean_d), ref_styles=npf(out['styles']),
                  ref_gen_imgs_sub2=npf(out['gen_imgs'][:, :, ::2, ::2]), ref_gen_thumb_imgs=npf(out['gen_thumb_imgs']),
                  ref_depth=npf(out['depth']), ref_plain_thumb=npf(plain['gen_thumb_imgs']))
    # ------------------------------------------------------------------ (2) base Generator.forward, surface-extraction shape
    gs = sm.Generator(syn.model_opt(size=256, channel_multiplier=1, renderer_spatial_output_dim=16),
                      syn.rendering_opt(N_samples=16), full_pipeline=False).eval()
    sd_s = {k: v for k, v in sd.items() if k in gs.state_dict()}
    missing, unexpected = gs.load_state_dict(sd_s, strict=False)
    assert not unexpected and all(k.endswith('.kernel') for k in missing), (missing, unexpected)
    wr, _ = syn.synthetic_inputs(1, seed=1)
    cs = cu.generate_camera_params(16, 'cpu', locations=torch.zeros(1, 2), fov_ang=6, dist_radius=0.12)
    with torch.no_grad():
        tup = gs([wr], cs[0], cs[1], cs[2], cs[3], input_is_latent=True, return_sdf=True, return_xyz=True)
    assert tup[0] is None and len(tup) == 5
    report['base_generator'] = dict(tuple_len=len(tup), shapes=[None if t is None else list(t.shape) for t in tup])
    arrays.update(s_poses=npf(cs[0]), s_focal=npf(cs[1]), s_near=npf(cs[2]), s_far=npf(cs[3]), s_styles_seed=np.int32(1),
                  s_ref_thumb=npf(tup[1]), s_ref_xyz=npf(tup[2]), s_ref_sdf=npf(tup[3]), s_ref_mask=npf(tup[4]))
    print(json.dumps(report, indent=1))
    with open(os.path.join(GOLD, "generator_z_base_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    save("generator_z_base", **arrays)


if __name__ == "__main__":
    main()
