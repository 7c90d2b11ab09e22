"""CPU restatement of the stage-1 training-direction outputs (eikonal terms via create_graph autograd, the 3-D
supervision re-queries) and the fixed scalar loss the gradient fixtures are recorded with.  TEST INFRASTRUCTURE: used by
oracle/gen_golden_grads.py and tests/ only."""
import torch

from . import renderer_ref


def stage1_loss(o, n_gt, g_feat):
    """Fixed scalar functional of the renderer outputs (shape of the stage-1 objective: eikonal_lambda 0.1,
    surface-normal L2, surface-sdf, uniform-points sdf 0.2, an image term and a feature term)."""
    return (0.1 * ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean() + (o['gen_thumb_imgs'] ** 2).mean()
            + ((o['xyz_rec_eikonal_term'] - n_gt) ** 2).mean() + (o['xyz_rec'] ** 2).mean()
            + 0.2 * (o['uniform_pts_rec'] ** 2).mean() + (o['features'] * g_feat).mean())


def restated(sd, poses, focal, near, far, styles, uni, surf, res, n_samples, dtype):
    """The same outputs from the oracle restatement, eikonal terms via autograd.grad(create_graph=True) as the
    reference builds them (:796-802)."""
    ro = renderer_ref.render(sd, poses, focal, near, far, styles, res=res, n_samples=n_samples, dtype=dtype)
    x = ro['points'].detach().clone().requires_grad_(True)
    raw = renderer_ref.query_points(sd, x, None, styles, dtype=dtype)
    eik = torch.autograd.grad(raw[..., 3:4], x, torch.ones_like(raw[..., 3:4]), create_graph=True)[0]
    xs = surf.to(dtype).unsqueeze(3).clone().requires_grad_(True)
    raw_s = renderer_ref.query_points(sd, xs, None, styles, dtype=dtype)
    eik_s = torch.autograd.grad(raw_s[..., 3:4], xs, torch.ones_like(raw_s[..., 3:4]), create_graph=True)[0]
    return dict(eikonal_term=eik, gen_thumb_imgs=ro['gen_thumb_imgs'], features=ro['features'],
                xyz_rec_eikonal_term=eik_s, xyz_rec=raw_s[..., 3:4],
                uniform_pts_rec=renderer_ref.query_points(sd, uni, None, styles, dtype=dtype)[..., 3:4])


def c5_loss(o):
    """The C5 loss of SURVEY.md 8d: mean(rgb^2) + mean((|eik| - 1)^2) + mean(surf_eik^2)."""
    return ((o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean()
            + (o['surface_eikonal_term'] ** 2).mean())


def restated_c5(sd, poses, focal, near, far, styles, res, n_samples, dtype):
    """render + eikonal term on the ray samples + the normal at the integrated surface point WITH the point kept in
    the graph (volume_integration :921-930: `xyz` is a function of the styles through the compositing weights), plus the
    compositing weights (hit_prob, read by cycle_runner.py:134)."""
    ro = renderer_ref.render(sd, poses, focal, near, far, styles, res=res, n_samples=n_samples, dtype=dtype)
    x = ro['points'].detach().clone().requires_grad_(True)
    raw = renderer_ref.query_points(sd, x, None, styles, dtype=dtype)
    eik = torch.autograd.grad(raw[..., 3:4], x, torch.ones_like(raw[..., 3:4]), create_graph=True)[0]
    xs = ro['xyz'].permute(0, 2, 3, 1).unsqueeze(3)                       # (B,H,W,1,3), attached
    if not xs.requires_grad:
        xs = xs.clone().requires_grad_(True)
    raw_s = renderer_ref.query_points(sd, xs, None, styles, dtype=dtype)
    se = torch.autograd.grad(raw_s[..., 3:4], xs, torch.ones_like(raw_s[..., 3:4]), create_graph=True)[0]
    return dict(eikonal_term=eik, surface_eikonal_term=se, gen_thumb_imgs=ro['gen_thumb_imgs'], features=ro['features'],
                hit_prob=ro['hit_prob'], xyz=ro['xyz'])
