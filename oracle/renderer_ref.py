"""CPU restatement of the StyleSDF volume renderer (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional PyTorch over a state dict with the reference's key names; works in float32 (the parity oracle and
the reported CPU baseline) and float64 (the "truth" both fp32 implementations are measured against).
Reference: project/utils/volume_renderer.py -- line numbers cited per function."""
import math

import torch
from torch.nn import functional as F

N_FILM = 9


def _w(sd, key, dtype):
    return sd[key].to(dtype)


def linear_layer(sd, prefix, x, std_init=1.0, bias_init=0.0):
    """LinearLayer.forward :76-80 -- the bias sits INSIDE the std_init scale."""
    dt = x.dtype
    return std_init * F.linear(x, _w(sd, prefix + 'weight', dt), _w(sd, prefix + 'bias', dt)) + bias_init


def film_siren(sd, prefix, h, style):
    """FiLMSiren.forward :116-132: sin(gamma(style) * (W h + b) + beta(style)); gamma = 15*lin+30, beta = 0.25*lin
    (:107-114).  h (B, ..., Cin), style (B, 256)."""
    dt = h.dtype
    out = F.linear(h, _w(sd, prefix + 'weight', dt), _w(sd, prefix + 'bias', dt))
    B = style.shape[0]
    shape = [B] + [1] * (h.ndim - 2) + [-1]
    gamma = linear_layer(sd, prefix + 'gamma.', style, 15.0, 30.0).reshape(shape)
    beta = linear_layer(sd, prefix + 'beta.', style, 0.25, 0.0).reshape(shape)
    return torch.sin(gamma * out + beta)


def film_params(sd, net_prefix, styles):
    """(B,9,2,256): the gamma / beta of all nine layers, for checking e3dge_film_params in isolation."""
    out = []
    for l in range(N_FILM):
        p = f'{net_prefix}pts_linears.{l}.' if l < 8 else f'{net_prefix}views_linears.'
        s = styles[:, l] if styles.ndim == 3 else styles
        out.append(torch.stack([linear_layer(sd, p + 'gamma.', s, 15.0, 30.0),
                                linear_layer(sd, p + 'beta.', s, 0.25, 0.0)], 1))
    return torch.stack(out, 1)


def siren_forward(sd, net_prefix, net_inputs, styles, tex=None):
    """SirenGenerator.forward :240-264 (+ forward_tex's per-point FiLM :217-220).
    net_inputs (B, ..., 6) = [warped xyz, viewdir]; styles (B,9,256) or (B,256); tex = (alpha, beta) or None.
    Returns (B, ..., 260) = [rgb3, sdf1, features256]."""
    pts, views = net_inputs[..., :3], net_inputs[..., 3:]
    h = pts
    for i in range(8):                                             # forward_generator :176-191
        s = styles[:, i] if styles.ndim == 3 else styles
        h = film_siren(sd, f'{net_prefix}pts_linears.{i}.', h, s)
    sdf = linear_layer(sd, net_prefix + 'sigma_linear.', h)        # forward_geo :206-208 (reads h BEFORE the tex FiLM)
    if tex is not None:
        alpha, beta = tex
        h = (alpha.to(h.dtype) + 1) * h + beta.to(h.dtype)         # :219-220
    hv = torch.cat([h, views], -1)                                 # :222
    s_view = styles[:, -1] if styles.ndim == 3 else styles         # :226-229
    feat = film_siren(sd, net_prefix + 'views_linears.', hv, s_view)
    rgb = linear_layer(sd, net_prefix + 'rgb_linear.', feat)       # :235
    return torch.cat([rgb, sdf, feat], -1)                         # :259-261


def get_rays(res, focal, c2w):
    """get_rays :769-794 with static_viewdirs.  focal (B,1,1), c2w (B,3,4) -> rays_o, rays_d, dirs (B,res,res,3)."""
    dt = c2w.dtype
    lin = torch.linspace(0.5, res - 0.5, res, dtype=torch.float32, device=c2w.device).to(dt)   # :666-670 (built in fp32 there)
    gi, gj = torch.meshgrid(lin, lin, indexing='ij')
    i = gi.t().unsqueeze(0)                                        # :672-674: i[r,c] = x of column c
    j = gj.t().unsqueeze(0)
    dirs = torch.stack([(i - res * .5) / focal, -(j - res * .5) / focal,
                        -torch.ones_like(i).expand(focal.shape[0], res, res)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:, None, None, :3, :3], -1)
    rays_o = c2w[:, None, None, :3, -1].expand(rays_d.shape)
    return rays_o, rays_d, dirs


def volume_integration(raw, z_vals, rays_d, pts, sigmoid_beta, force_background=True, mask_thresh=1.08):
    """volume_integration :809-943 for the sdf / return_xyz / force_background configuration."""
    dt = raw.dtype
    dists = z_vals[..., 1:] - z_vals[..., :-1]                     # :826
    norm = torch.norm(rays_d.unsqueeze(3), dim=-1)                 # :827-828  (B,H,W,1)
    dists = torch.cat([dists, torch.full_like(norm, 1e10)], -1) * norm   # :831-837
    rgb, sdf, feat = raw[..., :3], raw[..., 3:4], raw[..., 4:]
    sigma = torch.sigmoid(-sdf / sigmoid_beta) / sigmoid_beta      # :804-807, :853
    alpha = 1 - torch.exp(-sigma * dists.unsqueeze(-1))            # :860-861
    vis = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1, :]), 1. - alpha + 1e-10], 3), 3)[..., :-1, :]  # :869-876
    weights = alpha * vis                                          # :877
    if force_background:
        weights = torch.cat([weights[..., :-1, :], 1 - weights[..., :-1, :].sum(3, keepdim=True)], 3)   # :884-886
    rgb_map = -1 + 2 * torch.sum(weights * torch.sigmoid(rgb), 3)  # :888-890
    feat_map = torch.sum(weights * feat, 3)                        # :894
    xyz = torch.sum(weights * pts, 3)                              # :905
    depth = torch.sum(weights * z_vals.unsqueeze(-1), 3, keepdim=True)   # :907-909
    mask = (depth < mask_thresh).to(dt)                            # :910
    return dict(rgb_map=rgb_map, feat_map=feat_map, sdf=sdf, mask=mask, xyz=xyz, depth=depth, dists=dists,
                weights=weights)


def render(sd, c2w, focal, near, far, styles, res=64, n_samples=24, dist_radius=0.12, tex=None,
           prefix='renderer.', net_prefix=None, dtype=torch.float32, force_background=True):
    """VolumeFeatureRenderer.forward :1865-1972 -> render :1666-1701 -> render_rays :1183-1287 ->
    run_network :1052-1128, inference configuration.  Returns the reference's dict (same keys, shapes, layouts)."""
    if net_prefix is None:
        net_prefix = prefix + 'network.'
    c2w, focal, near, far, styles = [t.to(dtype) for t in (c2w, focal, near, far, styles)]
    B = c2w.shape[0]
    rays_o, rays_d, dirs = get_rays(res, focal.reshape(B, 1, 1), c2w)
    viewdirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)       # :1679
    t_vals = torch.linspace(0., 1. - 1 / n_samples, steps=n_samples, dtype=torch.float32, device=c2w.device).to(dtype).reshape(1, 1, 1, -1)  # :690-693
    nearb = near.reshape(B, 1, 1, 1) * torch.ones_like(rays_d[..., :1])
    farb = far.reshape(B, 1, 1, 1) * torch.ones_like(rays_d[..., :1])
    z_vals = nearb * (1. - t_vals) + farb * t_vals                 # :1211
    pts = rays_o.unsqueeze(3) + rays_d.unsqueeze(3) * z_vals.unsqueeze(-1)   # :1231-1233
    scale = 2 / (2 * dist_radius)                                  # :720
    net_in = torch.cat([pts * scale, viewdirs.unsqueeze(3).expand(pts.shape)], -1)   # :1074-1079
    raw = siren_forward(sd, net_prefix, net_in, styles, tex)
    vi = volume_integration(raw, z_vals, rays_d, pts, _w(sd, prefix + 'sigmoid_beta', dtype), force_background)
    return {
        'rays_o': rays_o, 'rays_d': rays_d, 'dists': vi['dists'], 'near': nearb, 'far': farb,
        'hit_prob': vi['weights'], 'surface_eikonal_term': None, 'points': pts, 'sdf': vi['sdf'],
        'gen_thumb_imgs': vi['rgb_map'].permute(0, 3, 1, 2).contiguous(),      # :1964
        'features': vi['feat_map'].permute(0, 3, 1, 2).contiguous(),           # :1967
        'mask': vi['mask'].permute(0, 4, 1, 2, 3).contiguous(),                # :1960
        'xyz': vi['xyz'].permute(0, 3, 1, 2).contiguous(),                     # :1958
        'eikonal_term': None, 'depth': vi['depth'], 'mesh': None, 'shading_mesh': None, 'debug_mesh': None,
        'viewdirs': viewdirs, 'raw': raw,
    }


def query_points(sd, pts, viewdirs, styles, dist_radius=0.12, prefix='renderer.', net_prefix=None,
                 dtype=torch.float32):
    """run_network :1052-1128 on an arbitrary (B, ..., 3) point set -> raw (B, ..., 260)."""
    if net_prefix is None:
        net_prefix = prefix + 'network.'
    pts, styles = pts.to(dtype), styles.to(dtype)
    vd = torch.zeros_like(pts) if viewdirs is None else viewdirs.to(dtype).expand(pts.shape)
    net_in = torch.cat([pts * (2 / (2 * dist_radius)), vd], -1)
    return siren_forward(sd, net_prefix, net_in, styles)


def tex_modulations(sd, prefix, feats, dtype=torch.float32):
    """ResnetBlockFC.forward (project/models/helper_modules/resnetfc.py:49-58) of
    netLocal.local_feat_to_tex_modulations_linear, then the split of SirenLocalGlobal.forward_backbone (:331-336).
    feats (.., cin) -> (alpha, beta), each (.., 256)."""
    x = feats.to(dtype)
    w = lambda k: _w(sd, prefix + k, dtype)
    net = F.linear(torch.relu(x), w('fc_0.weight'), w('fc_0.bias'))
    dx = F.linear(torch.relu(net), w('fc_1.weight'), w('fc_1.bias'))
    out = F.linear(x, w('shortcut.weight')) + dx
    return torch.split(out, 256, dim=-1)


def query_hitting_probability_fixed_interval(sd, wd_space_pts, ref_poses, ref_extrinsics, near, far, styles, n_samples,
                                             return_type='weights', dtype=torch.float32, prefix='renderer.'):
    """VolumeFeatureRenderer.query_hitting_probability_fixed_interval (project/utils/volume_renderer.py:1326-1495) with
    volume_integration(no_force_stop=True) (:826-837, :869-886).  wd_space_pts (B,H,W,S,3); near / far (B,H,W,1)."""
    B, H, W, S = wd_space_pts.shape[:4]
    Sn = n_samples
    pts = wd_space_pts.to(dtype).reshape(B, H * W, S, 3)
    t_vals = torch.linspace(0., 1. - 1 / Sn, steps=Sn).to(dtype).reshape(1, 1, 1, 1, Sn)     # :690-693
    near = near.to(dtype).reshape(B, H * W, 1, 1, 1)
    far = far.to(dtype).reshape(B, H * W, 1, 1, 1)
    w2c = torch.cat((ref_extrinsics.to(dtype), torch.zeros_like(ref_extrinsics[..., 0:1, :]).to(dtype)), dim=-2)
    w2c[..., -1, -1] = 1
    homo = torch.cat((pts, torch.ones_like(pts[..., 0:1])), dim=-1).unsqueeze(-1)            # B HW S 4 1
    ref_space = w2c.reshape(B, 1, 1, 4, 4) @ homo
    rays_d_ref = ref_space[..., :3, :] / (-ref_space[..., 2:3, :])                            # B HW S 3 1
    rays_d_wd = (ref_poses.to(dtype).reshape(B, 1, 1, 3, 4)[..., :3] @ rays_d_ref).permute(0, 1, 2, 4, 3)   # B HW S 1 3
    rays_o = ref_poses.to(dtype)[..., 3:4].permute(0, 2, 1).reshape(B, 1, 1, 1, 3)
    z_vals = near * (1. - t_vals) + far * t_vals                                              # B HW 1 1 S
    interval = (z_vals[..., 1:2] - z_vals[..., 0:1]) * rays_d_wd.norm(dim=-1, keepdim=True).permute(0, 1, 2, 4, 3)
    z_vals = z_vals.permute(0, 1, 2, 4, 3)                                                    # B HW 1 S 1
    q = rays_o + rays_d_wd * z_vals                                                           # B HW S S 3
    idx = (pts.unsqueeze(-2) - q[..., 0:1, :]).norm(dim=-1, keepdim=True) / interval + 1e-5   # B HW S 1 1
    lo = torch.clamp(idx.floor().long(), min=0, max=Sn - 1)
    hi = torch.clamp(idx.ceil().long(), min=0, max=Sn - 1)
    viewdirs = F.normalize(rays_d_ref.squeeze(-1), dim=-1)                                    # static_viewdirs
    out = torch.empty(B, H * W, S, 1, 1, dtype=dtype)
    beta = sd[prefix + 'sigmoid_beta'].to(dtype)
    for b in range(B):
        raw = query_points(sd, q[b:b + 1], viewdirs[b:b + 1].unsqueeze(3).expand(q[b:b + 1].shape), styles[b:b + 1], dtype=dtype)
        sdf = raw[..., 3:4]                                                                   # 1 HW S S 1
        zv = z_vals[b:b + 1].squeeze(-1)                                                      # 1 HW 1 S
        dists = zv[..., 1:] - zv[..., :-1]
        dists = torch.cat([dists, dists[..., 0:1]], -1)
        dists = dists * torch.norm(viewdirs[b:b + 1].unsqueeze(3), dim=-1)                    # :822-837 (norm of the unit view dirs)
        sigma = torch.sigmoid(-sdf / beta) / beta
        alpha = 1 - torch.exp(-sigma * dists.unsqueeze(-1))
        vis = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1, :]), 1. - alpha + 1e-10], -2), -2)[..., :-1, :]
        val = alpha * vis if return_type == 'weights' else vis
        f = torch.gather(val, 3, lo[b:b + 1])
        c = torch.gather(val, 3, hi[b:b + 1])
        out[b:b + 1] = torch.lerp(f, c, idx[b:b + 1] - lo[b:b + 1])
    return out.reshape(B, H, W, S, 1)
