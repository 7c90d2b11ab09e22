"""Golden vectors for the TRAINING direction, recorded from the REAL reference (imported via oracle/ref_harness.py) --
authoring container only.  TEST INFRASTRUCTURE.

    python oracle/gen_golden_grads.py      # writes tests/golden/grads_8x18.npz (+ grads_report.json)

One stage-1-shaped step on 8x8 rays x 18 samples: the reference's VolumeFeatureRenderer.forward
with return_eikonal, return_surface_eikonal and the 3-D supervision re-queries (uniform points, surface points with
normals), a fixed scalar loss over its outputs, loss.backward() to the W+ styles.  Stored: the inputs, the reference's
eikonal terms / re-query outputs / loss / dL/dstyles (`ref_*`) and the float64 evaluation of the restatement (`f64_*`).
The script also checks the restatement's own autograd (fp32) against the reference and prints the deviation."""
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
from oracle.training_ref import c5_loss, restated, restated_c5, stage1_loss  # noqa: E402
from oracle.gen_golden import build_reference_generator, maxdiff, npf, save  # noqa: E402

RES, S = 8, 18


def main():
    vr, sm, cu, op = ref_harness.modules()
    g, sd = build_reference_generator(sm, 256, 1, S, RES)
    wr, _ = syn.synthetic_inputs(1, seed=1)
    loc = torch.tensor([[0.25, -0.05]])
    c = cu.generate_camera_params(RES, 'cpu', locations=loc, fov_ang=6, dist_radius=0.12)
    poses, focal, near, far = c[0], c[1], c[2], c[3]
    rs = np.random.RandomState(11)
    uni = torch.from_numpy((0.12 * rs.uniform(-1, 1, (1, 96, 1, 1, 3))).astype(np.float32))
    surf = torch.from_numpy((0.08 * rs.uniform(-1, 1, (1, RES, RES, 3))).astype(np.float32))
    n_gt = torch.from_numpy(rs.normal(size=(1, RES, RES, 1, 3)).astype(np.float32))
    g_feat = torch.from_numpy(rs.normal(size=(1, 256, RES, RES)).astype(np.float32))

    styles = wr.clone().requires_grad_(True)
    out = g.renderer(poses, focal, near, far, styles=styles, return_eikonal=True, return_surface_eikonal=True,
                     geometry_sample={'uniform_pts': uni.clone(), 'xyz': surf.clone()})
    loss = stage1_loss(out, n_gt, g_feat)
    loss.backward()
    ref_grad = styles.grad.clone()

    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        s = wr.detach().to(dt).requires_grad_(True)
        o = restated(sd, poses, focal, near, far, s, uni, surf, RES, S, dt)
        l = stage1_loss(o, n_gt.to(dt), g_feat.to(dt))
        l.backward()
        res[name] = (o, l.detach(), s.grad)
    o32, l32, g32 = res["f32"]
    o64, l64, g64 = res["f64"]
    scale = float(ref_grad.abs().max())
    report = dict(
        restatement_vs_reference=dict(
            eikonal_term=maxdiff(out['eikonal_term'], o32['eikonal_term']),
            xyz_rec_eikonal_term=maxdiff(out['xyz_rec_eikonal_term'], o32['xyz_rec_eikonal_term']),
            uniform_pts_rec=maxdiff(out['uniform_pts_rec'], o32['uniform_pts_rec']),
            loss=abs(float(loss) - float(l32)), dstyles_rel=maxdiff(ref_grad, g32) / scale),
        reference_vs_f64=dict(
            eikonal_term=maxdiff(out['eikonal_term'], o64['eikonal_term']),
            xyz_rec_eikonal_term=maxdiff(out['xyz_rec_eikonal_term'], o64['xyz_rec_eikonal_term']),
            loss=abs(float(loss) - float(l64)), dstyles_rel=maxdiff(ref_grad, g64) / scale),
        dstyles_max_abs=scale)
    print(json.dumps(report, indent=1))
    with open(os.path.join(GOLD, "grads_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    save("grads_8x18", poses=npf(poses), focal=npf(focal), near=npf(near), far=npf(far), res=np.int32(RES),
         n_samples=np.int32(S), styles_seed=np.int32(1), uniform_pts=npf(uni), surface_pts=npf(surf), normals_gt=npf(n_gt),
         g_feat=npf(g_feat),
         ref_eikonal_term=npf(out['eikonal_term']), ref_surface_eikonal_term=npf(out['surface_eikonal_term']),
         ref_xyz_rec_eikonal_term=npf(out['xyz_rec_eikonal_term']), ref_xyz_rec=npf(out['xyz_rec']),
         ref_uniform_pts_rec=npf(out['uniform_pts_rec']), ref_loss=np.float64(float(loss)), ref_dstyles=npf(ref_grad),
         f64_eikonal_term=npf(o64['eikonal_term']), f64_xyz_rec_eikonal_term=npf(o64['xyz_rec_eikonal_term']),
         f64_loss=np.float64(float(l64)), f64_dstyles=g64.detach().numpy().astype(np.float64))

    main_c5(g, sd, cu)


def main_c5(g, sd, cu):
    """Second recorded step: SURVEY.md 8d's C5 loss -- mean(rgb^2) + mean((|eik|-1)^2) + mean(surf_eik^2) -- whose last term
    reaches the styles also through the integrated surface point (the reference keeps xyz in the graph, :921-930), and a
    third one that puts a weight on hit_prob (the compositing weights carry grad in the reference, cycle_runner.py:134)."""
    wr, _ = syn.synthetic_inputs(1, seed=1)
    c = cu.generate_camera_params(RES, 'cpu', locations=torch.tensor([[-0.15, 0.1]]), fov_ang=6, dist_radius=0.12)
    poses, focal, near, far = c[0], c[1], c[2], c[3]
    rs = np.random.RandomState(13)
    g_hit = torch.from_numpy(rs.normal(size=(1, RES, RES, S, 1)).astype(np.float32))

    def run_ref(loss_fn):
        styles = wr.clone().requires_grad_(True)
        out = g.renderer(poses, focal, near, far, styles=styles, return_eikonal=True, return_surface_eikonal=True)
        loss = loss_fn(out)
        loss.backward()
        return out, loss.detach(), styles.grad.clone()

    def run_mine(loss_fn, dt):
        s = wr.detach().to(dt).requires_grad_(True)
        o = restated_c5(sd, poses, focal, near, far, s, RES, S, dt)
        l = loss_fn(o)
        l.backward()
        return o, l.detach(), s.grad

    hit_loss = lambda o: c5_loss(o) + (o['hit_prob'] * g_hit.to(o['hit_prob'].dtype)).mean()
    out, loss, ref_grad = run_ref(c5_loss)
    _, loss_h, ref_grad_h = run_ref(hit_loss)
    o32, l32, g32 = run_mine(c5_loss, torch.float32)
    o64, l64, g64 = run_mine(c5_loss, torch.float64)
    _, l64h, g64h = run_mine(hit_loss, torch.float64)
    _, _, g32h = run_mine(hit_loss, torch.float32)
    # how much of the gradient travels through d xyz / d styles: the same loss with the surface point detached
    s = wr.detach().double().requires_grad_(True)
    from oracle import renderer_ref
    ro = renderer_ref.render(sd, poses, focal, near, far, s, res=RES, n_samples=S, dtype=torch.float64)
    xs = ro['xyz'].detach().permute(0, 2, 3, 1).unsqueeze(3).clone().requires_grad_(True)
    raw_s = renderer_ref.query_points(sd, xs, None, s, dtype=torch.float64)
    se = torch.autograd.grad(raw_s[..., 3:4], xs, torch.ones_like(raw_s[..., 3:4]), create_graph=True)[0]
    (se ** 2).mean().backward()
    g_det = s.grad.clone()
    s2 = wr.detach().double().requires_grad_(True)
    o2 = restated_c5(sd, poses, focal, near, far, s2, RES, S, torch.float64)
    (o2['surface_eikonal_term'] ** 2).mean().backward()
    scale = float(ref_grad.abs().max())
    report = dict(
        restatement_vs_reference=dict(surface_eikonal_term=maxdiff(out['surface_eikonal_term'], o32['surface_eikonal_term']),
                                      loss=abs(float(loss) - float(l32)), dstyles_rel=maxdiff(ref_grad, g32) / scale,
                                      dstyles_hit_rel=maxdiff(ref_grad_h, g32h) / float(ref_grad_h.abs().max())),
        reference_vs_f64=dict(surface_eikonal_term=maxdiff(out['surface_eikonal_term'], o64['surface_eikonal_term']),
                              loss=abs(float(loss) - float(l64)), dstyles_rel=maxdiff(ref_grad, g64) / scale,
                              dstyles_hit_rel=maxdiff(ref_grad_h, g64h) / float(ref_grad_h.abs().max())),
        surf_term_grad_through_xyz=dict(with_xyz_max=float(s2.grad.abs().max()), detached_max=float(g_det.abs().max()),
                                        difference_rel=float((s2.grad - g_det).abs().max() / s2.grad.abs().max())),
        dstyles_max_abs=scale)
    print(json.dumps(report, indent=1))
    with open(os.path.join(GOLD, "grads_c5_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    save("grads_c5_8x18", poses=npf(poses), focal=npf(focal), near=npf(near), far=npf(far), res=np.int32(RES),
         n_samples=np.int32(S), styles_seed=np.int32(1), g_hit=npf(g_hit),
         ref_eikonal_term=npf(out['eikonal_term']), ref_surface_eikonal_term=npf(out['surface_eikonal_term']),
         ref_loss=np.float64(float(loss)), ref_dstyles=npf(ref_grad), ref_loss_hit=np.float64(float(loss_h)),
         ref_dstyles_hit=npf(ref_grad_h), f64_surface_eikonal_term=npf(o64['surface_eikonal_term']),
         f64_loss=np.float64(float(l64)), f64_dstyles=g64.detach().numpy().astype(np.float64),
         f64_loss_hit=np.float64(float(l64h)), f64_dstyles_hit=g64h.detach().numpy().astype(np.float64),
         f64_dstyles_surf_only=s2.grad.detach().numpy().astype(np.float64),
         f64_dstyles_surf_only_detached_xyz=g_det.detach().numpy().astype(np.float64))


if __name__ == "__main__":
    main()
