"""CPU restatement of the per-point local-feature pipeline of the second renderer pass (TEST INFRASTRUCTURE -- see
oracle/__init__.py): projection + bilinear sampling + in-image masks + positional encoding + Fuse_sft_MLP, each function
citing the reference lines it follows."""
import torch
from torch.nn import functional as F


def perspective(points, calibs):
    """vendor/pifu/lib/geometry.py:101-129.  points (B,3,N), calibs (B,3|4,4) -> (B,3,N) = (x, y, depth)."""
    rot, trans = calibs[:, :3, :3], calibs[:, :3, 3:4]
    homo = torch.baddbmm(trans, rot, points)
    z = homo[:, 2:3, :] * -1 if homo[0, -1, 0] < 0 else homo[:, 2:3, :]
    return torch.cat([homo[:, :2, :] / z, z], 1)


def query(points, calibs, im_feat=None):
    """HGPIFuNetGAN.query (vendor/pifu/lib/model/HGPIFuGANNet.py:85-151) with return_projection_only / im_feat given.
    points (B,3,N) -> dict(proj_xy (B,2,N), depth (B,1,N), in_img (B,N) bool[, feats (B,C,N)])."""
    xyz = perspective(points, calibs)
    xyz[:, 1, :] = -1 * xyz[:, 1, :]
    xy, z = xyz[:, :2, :], xyz[:, 2:3, :]
    in_img = (xy[:, 0] >= -1.0) & (xy[:, 0] <= 1.0) & (xy[:, 1] >= -1.0) & (xy[:, 1] <= 1.0)
    out = dict(proj_xy=xy, depth=z, in_img=in_img)
    if im_feat is not None:                      # index(), geometry.py:64-80
        uv = xy.transpose(1, 2).unsqueeze(2)
        out['feats'] = F.grid_sample(im_feat, uv, mode='bilinear', padding_mode='zeros', align_corners=False)[:, :, :, 0]
    return out


def pos_encoding(x, n_freqs=7):
    """PosEncoding.forward (project/utils/misc_utils.py:148-185), logscale frequencies 2^0 .. 2^(n-1)."""
    out = [x]
    for f in 2 ** torch.linspace(0, n_freqs - 1, n_freqs):
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def _lin(sd, p, x):
    return F.linear(x, sd[p + 'weight'].to(x.dtype), sd[p + 'bias'].to(x.dtype) if p + 'bias' in sd else None)


def fuse_sft_mlp(sd, prefix, enc_feat, dec_feat, w=1):
    """Fuse_sft_MLP.forward (project/models/helper_modules/sft.py:103-109) with ResnetBlockFC (resnetfc.py:49-58)."""
    x = torch.cat([enc_feat, dec_feat], dim=-1)
    net = _lin(sd, prefix + 'encode_enc.fc_0.', torch.relu(x))
    dx = _lin(sd, prefix + 'encode_enc.fc_1.', torch.relu(net))
    e = _lin(sd, prefix + 'encode_enc.shortcut.', x) + dx
    mlp = lambda n: _lin(sd, f'{prefix}{n}.2.', F.leaky_relu(_lin(sd, f'{prefix}{n}.0.', e), 0.2))
    return dec_feat + w * (dec_feat * mlp('scale') + mlp('shift'))


def local_features(sd, prefix, pts5, xyz, ref_map, que_map, ref_calibs, que_calibs, n_freqs=7):
    """que_render_given_ref :212-300 (per-point part): (B,H,W,S,3) points -> (B,H,W,S,256+45) features, in_img mask."""
    B, H, W, S, _ = pts5.shape
    p = pts5.reshape(B, -1, 3).permute(0, 2, 1)
    q3 = query(p, ref_calibs, ref_map)
    f3 = q3['feats'].permute(0, 2, 1).reshape(B, H, W, S, -1)
    vis = query(xyz.reshape(B, 3, -1), ref_calibs)['in_img'].reshape(B, H, W, 1, 1).repeat_interleave(S, -2).to(pts5.dtype)
    f2 = query(p, que_calibs, que_map)['feats'].permute(0, 2, 1).reshape(B, H, W, S, -1)
    fused = fuse_sft_mlp(sd, prefix, torch.cat([f2, vis], -1), f3)
    return torch.cat((fused, pos_encoding(pts5, n_freqs)), -1), q3['in_img'].reshape(B, H, W, S, 1)
