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


def restated_stage2(sd, fuse_sd, head_sd, inp, pts5, xyz, ref_calibs, que_calibs, cam, styles, wd, res, n_samples, dtype,
                    fuse_prefix='Fuse_sft_block.', head_prefix='renderer.network.netLocal.local_feat_to_tex_modulations_linear.', g_feat=None):
    """One stage-2 pass (que_render_given_ref, e3dge_full_runner.py:185-317: query on both views -> Fuse_sft_MLP -> PE -> texture head
    -> second renderer pass with the per-point FiLM -> decoder) under autograd in `dtype`, with the loss of oracle/gen_golden_stage2.py:
    L = <g_img, image> + <g_rgb, thumbnail>.  fuse_sd / head_sd: un-prefixed parameter dicts; cam = (poses, focal, near, far).
    Returns (loss, image, thumbnail, {gradient name: tensor}) -- d_ref_map, d_que_map, d_styles, d_fuse.<param>, d_head.<param>.
    g_feat (B,256,H,W) given: the decoder is left out and L = <g_feat, feature map> + <g_rgb, thumbnail> (no lrelu' step functions
    between the loss and the path: the arithmetic can be held to fp32 accuracy); image is None then."""
    from oracle import decoder_ref, local_ref, renderer_ref
    leaf = lambda t: t.detach().to(dtype).clone().requires_grad_(True)
    d = lambda t: t.detach().to(dtype)
    rm, qm, w_ = leaf(inp['ref_map']), leaf(inp['que_map']), leaf(styles)
    fs = {fuse_prefix + k: leaf(v) for k, v in fuse_sd.items()}
    hs = {head_prefix + k: leaf(v) for k, v in head_sd.items()}
    f_, _ = local_ref.local_features(fs, fuse_prefix, d(pts5), d(xyz), rm, qm, d(ref_calibs), d(que_calibs))
    tex = renderer_ref.tex_modulations({**sd, **hs}, head_prefix, f_, dtype=dtype)
    o = renderer_ref.render(sd, cam[0], cam[1], cam[2], cam[3], w_, res=res, n_samples=n_samples, tex=tex, dtype=dtype)
    if g_feat is None:
        im = decoder_ref.decoder_forward(sd, o['features'], wd, noises=inp['noises'], dtype=dtype)
        loss = (im * d(inp['g_img'])).sum() + (o['gen_thumb_imgs'] * d(inp['g_rgb'])).sum()
    else:
        im = None
        loss = (o['features'] * d(g_feat)).sum() + (o['gen_thumb_imgs'] * d(inp['g_rgb'])).sum()
    gs = torch.autograd.grad(loss, [rm, qm, w_] + list(fs.values()) + list(hs.values()))
    grads = dict(d_ref_map=gs[0], d_que_map=gs[1], d_styles=gs[2])
    grads.update({'d_fuse.' + k[len(fuse_prefix):]: t for k, t in zip(fs, gs[3:3 + len(fs)])})
    grads.update({'d_head.' + k[len(head_prefix):]: t for k, t in zip(hs, gs[3 + len(fs):])})
    return float(loss.detach()), None if im is None else im.detach(), o["gen_thumb_imgs"].detach(), grads
