"""Second-pass local features from FEATURE MAPS -- host-side mirror of the per-point part of
`que_render_given_ref` (project/trainers/E3DGE/e3dge_full_runner.py:185-317):

    feature_3dprojection = netLocal.query(points, ref_calibs, im_feat=ref_map)['feats']          (:223-233)
    vis_mask             = netLocal.query(surface xyz, ref_calibs, projection only)['in_img']     (:246-255)
    feature_2dAlign      = netLocal.query(points, que_calibs, im_feat=que_map)['feats'] (+ mask)  (:275-292)
    fused                = Fuse_sft_MLP(feature_2dAlign, feature_3dprojection)                     (:295-296)
    feats                = cat(fused, PosEncoding(points))            -> (B,H,W,S,301)             (:299-300)
    (alpha, beta)        = netLocal.local_feat_to_tex_modulations_linear(feats)                    (volume_renderer.py:327-336)

What runs where: projection + bilinear gather + masks (e3dge_local_query) and the positional encoding (e3dge_pos_encoding)
are HIP kernels that write straight into the column slices of the buffers the MLPs read (no concatenation copies); the
texture head is the fused HIP kernel of round 1.  Fuse_sft_MLP (590 k MAC per point: a ResnetBlockFC(513 -> 256) and four
256x256 linears) is, for now, GPU library GEMMs through torch -- the next kernel to write.  The hourglass
image filters that PRODUCE the feature maps stay outside the path."""
import torch
from torch import nn

from . import _lib


def query_feature_map(pts, calibs, fmap=None, out=None, col_off=0, mask_out=None, mask_off=0, want_proj=False):
    """pts (B,N,3) world space, calibs (B,3,4), fmap (B,C,h,w) [any memory format; used channel-last] or None.
    Returns (feats view (B,N,C) or None, in_img (B,N) float 0/1, proj (B,N,3) or None).  With `out` (B,N,ld) the features
    are written into out[..., col_off:col_off+C]; with `mask_out` (B,N,ld') the mask goes to mask_out[..., mask_off]."""
    _lib.require_gpu(pts, "pts")
    _lib.require_gpu(calibs, "calibs")
    if torch.is_grad_enabled() and (pts.requires_grad or (fmap is not None and fmap.requires_grad)):
        # the reference's index() is grid_sample_gradfix: differentiable w.r.t. the feature map (stage-2 training trains the
        # hourglass filters through it).  The gather kernel has no backward yet -- refuse rather than drop the gradient.
        raise NotImplementedError("e3dge_local_query has no backward: detach the feature map / points (the hourglass filters "
                                  "that produce them are outside this build) or run under torch.no_grad()")
    B, N, _ = pts.shape
    dev = pts.device
    p = pts.contiguous()
    c = calibs[:, :3, :4].contiguous()
    C = h = w = 0
    fm = None
    if fmap is not None:
        _lib.require_gpu(fmap, "fmap")
        if fmap.shape[0] != B:
            raise RuntimeError(f"feature map batch {fmap.shape[0]} != points batch {B}")
        C, h, w = fmap.shape[1], fmap.shape[2], fmap.shape[3]
        fm = fmap.permute(0, 2, 3, 1).contiguous()                   # channel-last rows: a corner = C contiguous floats
        if out is None:
            out = torch.empty((B, N, C), device=dev, dtype=torch.float32)
            col_off = 0
    ld = out.shape[-1] if out is not None else 0
    if mask_out is None:
        mask = torch.empty((B, N), device=dev, dtype=torch.float32)
        m_ptr, m_ld, m_off = mask, 1, 0
    else:
        mask, m_ptr, m_ld, m_off = mask_out[..., mask_off], mask_out, mask_out.shape[-1], mask_off
    proj = torch.empty((B, N, 3), device=dev, dtype=torch.float32) if want_proj else None
    with torch.cuda.device(dev):
        rc = _lib.load().e3dge_local_query(_lib.ptr(out), ld, col_off, _lib.ptr(m_ptr), m_ld, m_off, _lib.ptr(proj), _lib.ptr(p),
                                           _lib.ptr(c), _lib.ptr(fm), B, N, C, h, w, _lib.stream_of(p))
    _lib.check(rc, "e3dge_local_query")
    feats = None if fmap is None else out[..., col_off:col_off + C]
    return feats, mask, proj


def pos_encoding(pts, n_freqs=7, out=None, col_off=0):
    """PosEncoding.forward (project/utils/misc_utils.py:148-185): (..., 3) -> (..., 3 * (2 n_freqs + 1)); with `out`
    (M, ld) the columns go to out[:, col_off:...]."""
    _lib.require_gpu(pts, "pts")
    lead = pts.shape[:-1]
    p = pts.reshape(-1, 3).contiguous()
    width = 3 * (2 * n_freqs + 1)
    own = out is None
    if own:
        out = torch.empty((p.shape[0], width), device=p.device, dtype=torch.float32)
        col_off = 0
    with torch.cuda.device(p.device):
        rc = _lib.load().e3dge_pos_encoding(_lib.ptr(out), out.shape[-1], col_off, _lib.ptr(p), p.shape[0], n_freqs, _lib.stream_of(p))
    _lib.check(rc, "e3dge_pos_encoding")
    return out.reshape(*lead, width) if own else out


class _ResnetBlockFCLib(nn.Module):
    """ResnetBlockFC (project/models/helper_modules/resnetfc.py:7-58) with library GEMMs; parameter names as there."""

    def __init__(self, size_in, size_out):
        super().__init__()
        size_h = min(size_in, size_out)
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = nn.Linear(size_in, size_out, bias=False) if size_in != size_out else None
        # the reference's initialisation (resnetfc.py:33-47): zero biases, kaiming fan-in weights, fc_1.weight zero
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)
        if self.shortcut is not None:
            nn.init.kaiming_normal_(self.shortcut.weight, a=0, mode="fan_in")

    def forward(self, x):
        net = self.fc_0(torch.relu(x))
        dx = self.fc_1(torch.relu(net))
        return (self.shortcut(x) if self.shortcut is not None else x) + dx


class Fuse_sft_MLP(nn.Module):
    """project/models/helper_modules/sft.py:84-109 (state-dict keys `encode_enc.*`, `scale.{0,2}.*`, `shift.{0,2}.*`)."""

    def __init__(self, in_ch=256 + 1, out_ch=256):
        super().__init__()
        self.encode_enc = _ResnetBlockFCLib(in_ch + out_ch, out_ch)
        self.scale = nn.Sequential(nn.Linear(out_ch, out_ch), nn.LeakyReLU(0.2, True), nn.Linear(out_ch, out_ch))
        self.shift = nn.Sequential(nn.Linear(out_ch, out_ch), nn.LeakyReLU(0.2, True), nn.Linear(out_ch, out_ch))

    def forward(self, enc_feat, dec_feat, w=1):
        return self.fuse(torch.cat([enc_feat, dec_feat], dim=-1), dec_feat, w)

    def fuse(self, enc_in, dec_feat, w=1):
        """enc_in = cat(enc_feat, dec_feat) already laid out in one buffer (the query kernels write it that way)."""
        e = self.encode_enc(enc_in)
        return dec_feat + w * (dec_feat * self.scale(e) + self.shift(e))


def local_features_from_maps(local_data_batch, n_freqs=7):
    """(B,H,W,S,301) per-point local features from the two feature maps (see the module docstring).  Keys of
    `local_data_batch`: 'feature_maps' = {'ref': (B,C,h,w), 'que': (B,C,h,w)}, 'ref_calibs', 'que_calibs' (B,3,4),
    'points' (B,H,W,S,3) world-space samples of the query view, 'xyz' (B,3,H,W) its integrated surface points,
    'fuse_sft_block' (Fuse_sft_MLP), optional 'add_vis_mask' (default True)."""
    maps = local_data_batch['feature_maps']
    pts5 = local_data_batch['points']
    B, H, W, S, _ = pts5.shape
    N = H * W * S
    pts = pts5.reshape(B, N, 3)
    fuse = local_data_batch['fuse_sft_block']
    add_mask = bool(local_data_batch.get('add_vis_mask', True))
    C = maps['ref'].shape[1]
    n_enc = C + (1 if add_mask else 0)
    enc_in = torch.empty((B, N, n_enc + C), device=pts.device, dtype=torch.float32)       # [2D-aligned | vis mask | 3D-projected]
    query_feature_map(pts, local_data_batch['que_calibs'], maps['que'], out=enc_in, col_off=0)
    dec, in_img, _ = query_feature_map(pts, local_data_batch['ref_calibs'], maps['ref'], out=enc_in, col_off=n_enc)
    if add_mask:   # is the query view's surface point inside the reference image? one value per ray, shared by its samples
        surf = local_data_batch['xyz'].reshape(B, 3, H * W).permute(0, 2, 1)
        _, vis, _ = query_feature_map(surf, local_data_batch['ref_calibs'])
        enc_in[..., C] = vis.reshape(B, H * W, 1).expand(B, H * W, S).reshape(B, N)
    width = 3 * (2 * n_freqs + 1)
    feats = torch.empty((B, N, C + width), device=pts.device, dtype=torch.float32)
    feats[..., :C] = fuse.fuse(enc_in, dec)
    pos_encoding(pts, n_freqs, out=feats.reshape(B * N, C + width), col_off=C)
    return feats.reshape(B, H, W, S, C + width), in_img.reshape(B, H, W, S, 1)


def tex_modulations_from_maps(local_head, renderer, cam_poses, focal, near, far, local_data_batch):
    """(alpha, beta), each (B,H,W,S,256), for the renderer's second pass (called by VolumeFeatureRenderer.forward)."""
    feats, in_img = local_features_from_maps(local_data_batch)
    local_data_batch['in_img_mask'] = in_img
    return local_head.local_feat_to_tex_modulations_linear.tex_modulations(feats)
