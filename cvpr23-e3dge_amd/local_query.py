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
256x256 linears) runs as nine weight-stationary split-f16 launches (e3dge_ws_linear, round 3) when no autograd graph is
needed, as torch modules (library GEMMs) otherwise.  The hourglass image filters that PRODUCE the feature maps stay outside the
path."""
import ctypes
import os
import weakref

import torch
from torch import nn

from . import _lib

_FUSE_IMAGES = weakref.WeakKeyDictionary()          # Fuse_sft_MLP -> packed weight images of its nine 256 x 256 blocks


def fuse_autograd_backend():
    """'hip' (default): Fuse_sft_MLP under autograd runs its forward as the nine e3dge_ws_linear launches (_FuseFn);
    E3DGE_FUSE_AUTOGRAD=torch keeps the torch modules there (A/B, tests)."""
    v = os.environ.get("E3DGE_FUSE_AUTOGRAD", "hip")
    if v not in ("hip", "torch"):
        raise RuntimeError(f"E3DGE_FUSE_AUTOGRAD must be 'hip' or 'torch', got {v!r}")
    return v


def native_fuse_backend():
    """'hip' (default): Fuse_sft_MLP without an autograd graph runs as nine e3dge_ws_linear launches; E3DGE_FUSE=torch keeps
    the library-GEMM modules (A/B, tests)."""
    v = os.environ.get("E3DGE_FUSE", "hip")
    if v not in ("hip", "torch"):
        raise RuntimeError(f"E3DGE_FUSE must be 'hip' or 'torch', got {v!r}")
    return v


def query_feature_map(pts, calibs, fmap=None, out=None, col_off=0, mask_out=None, mask_off=0, want_proj=False):
    """pts (B,N,3) world space, calibs (B,3,4), fmap (B,C,h,w) [any memory format; used channel-last] or None.
    Returns (feats view (B,N,C) or None, in_img (B,N) float 0/1, proj (B,N,3) or None).  With `out` (B,N,ld) the features
    are written into out[..., col_off:col_off+C]; with `mask_out` (B,N,ld') the mask goes to mask_out[..., mask_off]."""
    _lib.require_gpu(pts, "pts")
    _lib.require_gpu(calibs, "calibs")
    if torch.is_grad_enabled() and (pts.requires_grad or (fmap is not None and fmap.requires_grad)):
        # the reference's index() is grid_sample_gradfix: differentiable w.r.t. the feature map AND the sampling position (stage-2
        # training trains the hourglass filters through it).  The differentiable form returns its own (B,N,C) tensor:
        if fmap is None or want_proj or out is not None or mask_out is not None:
            raise NotImplementedError("the differentiable gather (e3dge_local_query_bwd) covers the features only: call without "
                                      "`out` / `mask_out` / want_proj, or detach the points (projection and mask carry no gradient here)")
        feats, mask = _GatherFn.apply(pts, calibs, fmap)
        return feats, mask, None
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


class _GatherFn(torch.autograd.Function):
    """e3dge_local_query with a backward (e3dge_local_query_bwd): bilinear scatter of d feats into the channel-last map (atomic
    adds) and the gradient through the sampling position; the in-image mask is returned as a non-differentiable output.
    Reference: project/models/op/grid_sample_gradfix.py:52-89."""

    @staticmethod
    def forward(ctx, pts, calibs, fmap):
        with torch.no_grad():
            feats, mask, _ = query_feature_map(pts.detach(), calibs.detach(), fmap.detach())
        ctx.save_for_backward(pts, calibs, fmap)
        ctx.mark_non_differentiable(mask)
        return feats, mask

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_feats, _d_mask):
        pts, calibs, fmap = ctx.saved_tensors
        B, N, _ = pts.shape
        C, h, w = fmap.shape[1:]
        dev = pts.device
        need_p, _, need_f = ctx.needs_input_grad
        g = d_feats.contiguous().float()
        fm = fmap.detach().permute(0, 2, 3, 1).contiguous()
        d_fm = torch.zeros((B, h, w, C), device=dev, dtype=torch.float32) if need_f else None
        d_p = torch.empty((B, N, 3), device=dev, dtype=torch.float32) if need_p else None
        p = pts.detach().contiguous()
        c = calibs.detach()[:, :3, :4].contiguous()
        with torch.cuda.device(dev):
            rc = _lib.load().e3dge_local_query_bwd(_lib.ptr(d_fm), _lib.ptr(d_p), _lib.ptr(g), C, 0, _lib.ptr(p), _lib.ptr(c), _lib.ptr(fm),
                                                   B, N, C, h, w, _lib.stream_of(g))
        _lib.check(rc, "e3dge_local_query_bwd")
        return d_p, None, (d_fm.permute(0, 3, 1, 2) if need_f else None)


class _EncInFn(torch.autograd.Function):
    """Round 6: the 2 C (+ 1)-wide input rows of Fuse_sft_MLP -- [gather on the query view's map | visibility mask | gather on the reference
    view's map] (e3dge_full_runner.py:185-317, HGPIFuGANNet.py:85-151) -- as ONE node: both e3dge_local_query launches write into the row
    buffer (no torch.cat), and the backward hands each e3dge_local_query_bwd its column block of the incoming gradient in place (row pitch
    and offset; no .contiguous() of the slices).  Gradient to the two feature maps only: points that require grad take the composed path."""

    @staticmethod
    def forward(ctx, pts, que_calibs, ref_calibs, vis, que_map, ref_map):
        B, N, _ = pts.shape
        C = que_map.shape[1]
        n_enc = C + (1 if vis is not None else 0)
        with torch.no_grad():
            enc_in = torch.empty((B, N, n_enc + C), device=pts.device, dtype=torch.float32)
            query_feature_map(pts.detach(), que_calibs.detach(), que_map.detach(), out=enc_in, col_off=0)
            _, in_img, _ = query_feature_map(pts.detach(), ref_calibs.detach(), ref_map.detach(), out=enc_in, col_off=n_enc)
            if vis is not None:
                enc_in[..., C] = vis
        # (the backward never reads the maps' VALUES -- d map is a scatter of the incoming rows, and no gradient goes to the points here --
        # so it needs neither the maps nor their channel-last copies: 25 us per 16-MB transposing copy, twice per step, gone)
        ctx.map_shapes = (tuple(que_map.shape), tuple(ref_map.shape))
        ctx.save_for_backward(pts, que_calibs, ref_calibs)
        ctx.n_enc = n_enc
        ctx.mark_non_differentiable(in_img)
        return enc_in, in_img

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_enc, _d_mask):
        pts, que_calibs, ref_calibs = ctx.saved_tensors
        B, N, _ = pts.shape
        g = d_enc.float()
        if not g.is_contiguous():
            g = g.contiguous()
        ld = g.shape[-1]
        p = pts.detach().contiguous()
        out = []
        lib = _lib.load()
        sort_ref = os.environ.get("E3DGE_GATHER_BWD_SORT", "1") != "0"
        # (the points are samples along the QUERY view's rays: in that view's map a ray is one pixel and the kernel's run accumulation merges
        # its samples; in the reference view's map every sample lands on its own pixel -- those are walked in pixel order instead)
        for need, shape, calibs, off, other_view in ((ctx.needs_input_grad[4], ctx.map_shapes[0], que_calibs, 0, False),
                                                     (ctx.needs_input_grad[5], ctx.map_shapes[1], ref_calibs, ctx.n_enc, True)):
            if not need:
                out.append(None)
                continue
            C, h, w = shape[1:]
            d_fm = torch.zeros((B, h, w, C), device=g.device, dtype=torch.float32)
            fm = d_fm                                            # (fmap_nhwc is only read for d pts, which this node does not produce; any valid pointer)
            c = calibs.detach()[:, :3, :4].contiguous()
            with torch.cuda.device(g.device):
                if other_view and sort_ref and B * N < 2 ** 31:
                    n_ws = lib.e3dge_local_query_sort_ws_ints(B, N, h, w)
                    ws = torch.empty(n_ws, device=g.device, dtype=torch.int32)
                    rc = lib.e3dge_local_query_bwd_sorted(_lib.ptr(d_fm), None, _lib.ptr(g), ld, off, _lib.ptr(p), _lib.ptr(c), _lib.ptr(fm),
                                                          B, N, C, h, w, _lib.ptr(ws), n_ws, _lib.stream_of(g))
                else:
                    rc = lib.e3dge_local_query_bwd(_lib.ptr(d_fm), None, _lib.ptr(g), ld, off, _lib.ptr(p), _lib.ptr(c), _lib.ptr(fm),
                                                   B, N, C, h, w, _lib.stream_of(g))
            _lib.check(rc, "e3dge_local_query_bwd")
            out.append(d_fm.permute(0, 3, 1, 2))
        return None, None, None, None, out[0], out[1]


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

    def fuse(self, enc_in, dec_feat, w=1, out=None, out_off=0):
        """enc_in = cat(enc_feat, dec_feat) already laid out in one buffer (the query kernels write it that way).  With `out`
        (..., ld) the result goes to out[..., out_off:out_off + out_ch] (and that view is returned).
        GPU, fp32: nine launches of e3dge_ws_linear (weight-stationary split-f16, csrc/siren_ws.hip) -- under autograd as the forward
        of _FuseFn (E3DGE_FUSE_AUTOGRAD=torch, a non-GPU tensor or an `out` buffer under autograd: the torch modules)."""
        if self._native_ok(enc_in):
            if not self._wants_grad(enc_in):
                return self._fuse_native(enc_in, float(w), out, out_off)
            if out is None and fuse_autograd_backend() == "hip" and self._fusefn_ok(enc_in, dec_feat, w):
                # training (round 4): the same nine launches as the forward of an autograd node; its backward is library GEMMs on
                # the intermediates the launches left behind (the 3D-projected block of enc_in IS dec_feat, as on the inference path)
                return _FuseFn.apply(self, enc_in, float(w), None, 0, False, *self._param_list())
        e = self.encode_enc(enc_in)
        res = dec_feat + w * (dec_feat * self.scale(e) + self.shift(e))
        if out is None:
            return res
        out[..., out_off:out_off + res.shape[-1]] = res
        return out[..., out_off:out_off + res.shape[-1]]

    # ---- native path ----------------------------------------------------------------------------------------------------
    def _native_ok(self, enc_in):
        if native_fuse_backend() != "hip" or enc_in.device.type != "cuda" or enc_in.dtype != torch.float32:
            return False
        fc0 = self.encode_enc.fc_0
        if fc0.out_features != 256 or self.encode_enc.fc_1.out_features != 256 or fc0.in_features not in (512, 513):
            return False
        if enc_in.shape[-1] != fc0.in_features or self.encode_enc.shortcut is None:
            return False
        return all(p.device == enc_in.device and p.dtype == torch.float32 for p in _lib.params_of(self))

    def _fusefn_ok(self, enc_in, dec_feat, w):
        """_FuseFn differentiates w.r.t. enc_in and the parameters only, and reads dec_feat out of enc_in's last 256 columns: an empty
        input, a weight tensor that requires grad, or a dec_feat that is NOT that view (its own gradient would be dropped) take the torch
        modules instead (round-4 advisor findings)."""
        if enc_in.numel() == 0 or (torch.is_tensor(w) and (w.requires_grad or w.numel() != 1)):
            return False
        b_off = enc_in.shape[-1] - 256
        tail = enc_in[..., b_off:]
        return (torch.is_tensor(dec_feat) and dec_feat.shape == tail.shape and dec_feat.data_ptr() == tail.data_ptr()
                and dec_feat.stride() == tail.stride())

    def _wants_grad(self, enc_in):
        return torch.is_grad_enabled() and (enc_in.requires_grad or any(p.requires_grad for p in _lib.params_of(self)))

    def _param_list(self):
        """The thirteen parameters in the order _FuseFn.backward returns their gradients."""
        e = self.encode_enc
        return [e.fc_0.weight, e.fc_0.bias, e.fc_1.weight, e.fc_1.bias, e.shortcut.weight,
                self.scale[0].weight, self.scale[0].bias, self.scale[2].weight, self.scale[2].bias,
                self.shift[0].weight, self.shift[0].bias, self.shift[2].weight, self.shift[2].bias]

    def _images(self, device):
        """Packed weight images (e3dge_ws_pack) of the nine 256 x 256 blocks, bias / mask-column vectors; rebuilt when a
        parameter changes.  Kept outside the module (weak map): modules stay deep-copyable and state_dict-clean."""
        key = _lib.param_key(self) + (str(device),)
        hit = _FUSE_IMAGES.get(self)
        if hit is not None and hit['key'] == key:
            return hit
        lib = _lib.load()
        enc, n_in = self.encode_enc, self.encode_enc.fc_0.in_features
        has_col = n_in == 513
        b_off = 257 if has_col else 256                       # first column of the 3D-projected (dec) block

        def img(w):
            w = w.detach().contiguous()
            t = torch.empty(lib.e3dge_ws_image_bytes(1), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _lib.check(lib.e3dge_ws_pack(_lib.ptr(t), _lib.ptr(w), 1, _lib.stream_of(w)), "e3dge_ws_pack")
            return t
        f0, sc = enc.fc_0.weight, enc.shortcut.weight
        hit = dict(key=key, b_off=b_off, has_col=has_col,
                   f0a=img(f0[:, :256]), f0b=img(f0[:, b_off:]), f1=img(enc.fc_1.weight), sa=img(sc[:, :256]), sb=img(sc[:, b_off:]),
                   sc1=img(self.scale[0].weight), sc2=img(self.scale[2].weight), sh1=img(self.shift[0].weight), sh2=img(self.shift[2].weight),
                   f0col=f0[:, 256].detach().contiguous() if has_col else None, scol=sc[:, 256].detach().contiguous() if has_col else None,
                   b0=enc.fc_0.bias.detach().contiguous(), b1=enc.fc_1.bias.detach().contiguous(),
                   bsc1=self.scale[0].bias.detach().contiguous(), bsc2=self.scale[2].bias.detach().contiguous(),
                   bsh1=self.shift[0].bias.detach().contiguous(), bsh2=self.shift[2].bias.detach().contiguous(),
                   slope=float(self.scale[1].negative_slope))
        _FUSE_IMAGES[self] = hit
        return hit

    def _images_t(self, device):
        """Packed images of the TRANSPOSED 256 x 256 blocks (d input = d output @ W is the layer of W^T): built on the first backward."""
        I = self._images(device)
        if 'sc2_t' not in I:
            lib = _lib.load()
            enc, b_off = self.encode_enc, I['b_off']

            def img_t(w):
                w = w.detach().t().contiguous()
                t = torch.empty(lib.e3dge_ws_image_bytes(1), dtype=torch.uint8, device=device)
                with torch.cuda.device(device):
                    _lib.check(lib.e3dge_ws_pack(_lib.ptr(t), _lib.ptr(w), 1, _lib.stream_of(w)), "e3dge_ws_pack")
                return t
            f0, sc = enc.fc_0.weight, enc.shortcut.weight
            I.update(sc2_t=img_t(self.scale[2].weight), sh2_t=img_t(self.shift[2].weight), sc1_t=img_t(self.scale[0].weight),
                     sh1_t=img_t(self.shift[0].weight), f1_t=img_t(enc.fc_1.weight), sa_t=img_t(sc[:, :256]), sb_t=img_t(sc[:, b_off:]),
                     f0a_t=img_t(f0[:, :256]), f0b_t=img_t(f0[:, b_off:]))
        return I

    def _fuse_bwd_native(self, g, x, net, s1, t1, scale, am_x, w, b_off, slope, need_x, ld_g=256, mask_col=True):
        """The data-gradient chain of sft.py:84-109 + resnetfc.py:49-58 as e3dge_ws_linear launches on the transposed images (round 5):
        dz1 = (w g . dec) Wsc2 . lrelu'(s1), dz2 = (w g) Wsh2 . lrelu'(t1), de = dz1 Wsc1 + dz2 Wsh1, dnet = de W1 . [net > 0],
        dx = de Ws + (dnet W0) . [x > 0], dx[dec block] += g (1 + w scale).  Returns (dz1, dz2, de, dnet, dx or None, the amax buffers of g / dz1 / dz2 / de / dnet).
        `g`: (N, 256) rows of pitch `ld_g` floats (round 6: the first 256 of the 301 gradient columns of the assembled features, read in place)."""
        N = g.shape[0]
        dev = g.device
        ld = x.shape[1]
        I = self._images_t(dev)
        lib = _lib.load()
        st = _lib.stream_of(g)
        f32 = dict(device=dev, dtype=torch.float32)
        dz1, dz2, de, dnet = (torch.empty((N, 256), **f32) for _ in range(4))
        dx = torch.empty((N, ld), **f32) if need_x else None
        am = torch.zeros((5, _lib.AMAX_FLOATS), **f32)              # g, dz1, dz2, de, dnet

        def lin(wimg, xin, am_in, y, ld_x=256, ld_y=256, off_y=0, post=0, sl=0.0, r1=None, r1_ld=256, r1_off=0, r2=None, r2_ld=256, r2_off=0,
                xmul=None, amax_out=None, x_scale=1.0):
            a = _lib.WsLinear()
            a.wimg, a.x, a.amax_in, a.y, a.amax_out = _lib.ptr(wimg), _lib.ptr(xin), _lib.ptr(am_in), _lib.ptr(y), _lib.ptr(amax_out)
            a.r1, a.r2 = _lib.ptr(r1), _lib.ptr(r2)
            a.n_rows = N
            a.ld_x, a.off_x, a.ld_y, a.off_y = ld_x, 0, ld_y, off_y
            a.ld_r1, a.off_r1, a.ld_r2, a.off_r2, a.ld_m, a.off_m = r1_ld, r1_off, r2_ld, r2_off, 1, 0
            a.post, a.slope, a.w_fuse, a.x_scale = post, sl, w, x_scale
            if xmul is not None:
                a.xmul, a.amax_xmul, a.ld_xmul, a.off_xmul = _lib.ptr(xmul), _lib.ptr(am_x), ld, b_off
            _lib.check(lib.e3dge_ws_linear(ctypes.byref(a), st), "e3dge_ws_linear")
        with torch.cuda.device(dev):
            if ld_g == 256:
                _lib.check(lib.e3dge_amax(_lib.ptr(am[0]), _lib.ptr(g), g.numel(), st), "e3dge_amax")
            else:
                _lib.check(lib.e3dge_amax_rows(_lib.ptr(am[0]), _lib.ptr(g), N, 256, ld_g, st), "e3dge_amax_rows")
            lin(I['sc2_t'], g, am[0], dz1, ld_x=ld_g, post=3, sl=slope, r1=s1, xmul=x, x_scale=w, amax_out=am[1])
            lin(I['sh2_t'], g, am[0], dz2, ld_x=ld_g, post=3, sl=slope, r1=t1, x_scale=w, amax_out=am[2])
            lin(I['sc1_t'], dz1, am[1], de)
            lin(I['sh1_t'], dz2, am[2], de, r1=de, amax_out=am[3])
            lin(I['f1_t'], de, am[3], dnet, post=3, sl=0.0, r1=net, amax_out=am[4])
            if need_x:
                lin(I['sa_t'], de, am[3], dx, ld_y=ld, off_y=0)
                lin(I['f0a_t'], dnet, am[4], dx, ld_y=ld, off_y=0, post=3, sl=0.0, r1=x, r1_ld=ld, r1_off=0, r2=dx, r2_ld=ld, r2_off=0)
                lin(I['sb_t'], de, am[3], dx, ld_y=ld, off_y=b_off, post=4, r1=g, r1_ld=ld_g, r2=scale)
                lin(I['f0b_t'], dnet, am[4], dx, ld_y=ld, off_y=b_off, post=3, sl=0.0, r1=x, r1_ld=ld, r1_off=b_off, r2=dx, r2_ld=ld, r2_off=b_off)
            if need_x and I['has_col'] and not mask_col:
                dx[:, 256].zero_()
            if need_x and I['has_col'] and mask_col:
                # the visibility-mask column (one input column of fc_0 and of the shortcut): two row dot products in one launch
                _lib.check(lib.e3dge_ws_rowdot2(_lib.ptr(dx), ld, 256, _lib.ptr(de), _lib.ptr(I['scol']), _lib.ptr(dnet), _lib.ptr(I['f0col']),
                                                _lib.ptr(x), ld, 256, N, st), "e3dge_ws_rowdot2")
        return dz1, dz2, de, dnet, dx, am

    def _fuse_native(self, enc_in, w, out, out_off, keep=None):
        """`keep` (a dict): every intermediate gets a buffer of its own and is left there for _FuseFn.backward --
        net (fc_0's output), e (the block's output), s1 / t1 (the SFT branches' hidden activations), scale (the scale branch)."""
        lead = enc_in.shape[:-1]
        x = enc_in.reshape(-1, enc_in.shape[-1])
        if not x.is_contiguous():
            x = x.contiguous()
        N, ld = x.shape
        dev = x.device
        I = self._images(dev)
        if out is None:
            out = torch.empty(lead + (256,), device=dev, dtype=torch.float32)
            out_off = 0
        o2 = out.reshape(-1, out.shape[-1])
        if o2.data_ptr() != out.data_ptr() or not o2.is_contiguous():
            raise RuntimeError("Fuse_sft_MLP.fuse: `out` must be a contiguous (..., ld) buffer")
        if N == 0:
            return out[..., out_off:out_off + 256]
        lib = _lib.load()
        A, Bf, C = (torch.empty((N, 256), device=dev, dtype=torch.float32) for _ in range(3))
        if keep is not None:
            NET, E, S1, T1 = (torch.empty((N, 256), device=dev, dtype=torch.float32) for _ in range(4))
        else:
            NET, E, S1, T1 = Bf, Bf, A, A
        am = torch.zeros((5, _lib.AMAX_FLOATS), device=dev, dtype=torch.float32)       # x, net, e, h1, h2
        st = _lib.stream_of(x)
        b_off = I['b_off']

        def lin(wimg, xin, ld_x, off_x, amax_in, y, ld_y=256, off_y=0, bias=None, col=None, r1=None, r1_ld=256, r1_off=0, r2=None,
                pre_relu=False, post=0, amax_out=None):
            a = _lib.WsLinear()
            a.wimg, a.x, a.amax_in, a.bias = _lib.ptr(wimg), _lib.ptr(xin), _lib.ptr(amax_in), _lib.ptr(bias)
            a.colw, a.m = (_lib.ptr(col), _lib.ptr(x)) if col is not None else (None, None)
            a.r1, a.r2, a.y, a.amax_out = _lib.ptr(r1), _lib.ptr(r2), _lib.ptr(y), _lib.ptr(amax_out)
            a.n_rows = N
            a.ld_x, a.off_x, a.ld_m, a.off_m = ld_x, off_x, ld, 256
            a.ld_r1, a.off_r1, a.ld_r2, a.off_r2, a.ld_y, a.off_y = r1_ld, r1_off, 256, 0, ld_y, off_y
            a.pre_relu, a.post, a.slope, a.w_fuse = int(pre_relu), post, I['slope'], w
            _lib.check(lib.e3dge_ws_linear(ctypes.byref(a), st), "e3dge_ws_linear")
        with torch.cuda.device(dev):
            _lib.check(lib.e3dge_amax(_lib.ptr(am[0]), _lib.ptr(x), x.numel(), st), "e3dge_amax")
            # ResnetBlockFC: net = fc_0(relu(x)), dx = fc_1(relu(net)), e = shortcut(x) + dx  (K = 513 as two 256-blocks + the mask column)
            lin(I['f0a'], x, ld, 0, am[0], A, pre_relu=True)
            lin(I['f0b'], x, ld, b_off, am[0], NET, bias=I['b0'], col=I['f0col'], r1=A, pre_relu=True, amax_out=am[1])
            lin(I['f1'], NET, 256, 0, am[1], A, bias=I['b1'], pre_relu=True)
            lin(I['sa'], x, ld, 0, am[0], C)
            lin(I['sb'], x, ld, b_off, am[0], E, col=I['scol'], r1=C, r2=A, amax_out=am[2])
            # SFT branches on e, then  dec + w (dec * scale + shift)
            lin(I['sc1'], E, 256, 0, am[2], S1, bias=I['bsc1'], post=1, amax_out=am[3])
            lin(I['sc2'], S1, 256, 0, am[3], C, bias=I['bsc2'])
            lin(I['sh1'], E, 256, 0, am[2], T1, bias=I['bsh1'], post=1, amax_out=am[4])
            lin(I['sh2'], T1, 256, 0, am[4], o2, ld_y=o2.shape[-1], off_y=out_off, bias=I['bsh2'], r1=x, r1_ld=ld, r1_off=b_off, r2=C, post=2)
        if keep is not None:
            keep.update(x=x, net=NET, e=E, s1=S1, t1=T1, scale=C, b_off=b_off, slope=I['slope'], am_x=am[0], am=am)
        return out[..., out_off:out_off + 256]


class _FuseFn(torch.autograd.Function):
    """Fuse_sft_MLP under autograd: forward = the nine weight-stationary launches of the inference path, keeping net / e / the two hidden
    SFT activations / the scale branch; backward = the chain rule of sft.py:84-109 + resnetfc.py:49-58 written out on those -- since
    round 5 the data-gradient chain is nine more e3dge_ws_linear launches on the transposed weight images (Fuse_sft_MLP._fuse_bwd_native;
    E3DGE_FUSE_BWD=torch keeps round 4's library GEMMs), the thirteen parameter gradients stay library GEMMs.  Not double-differentiable."""

    @staticmethod
    def forward(ctx, mod, enc_in, w, pe_pts, n_freqs, mask_col_const, *params):
        """pe_pts (round 6): None, or the points (.., 3) whose positional encoding (PosEncoding.forward, misc_utils.py:148-185) fills the
        columns behind the 256 fused ones -- the node then returns the assembled (.., 256 + 3 (2 n_freqs + 1)) feature rows, written in place
        (no torch.cat), and its backward reads the first 256 gradient columns in place (no .contiguous()).
        mask_col_const: the caller guarantees that nothing differentiates through column 256 of a 513-wide enc_in (the visibility mask written
        by _EncInFn): its gradient is returned as zeros instead of the two row dot products of e3dge_ws_rowdot2 (50 us at 98,304 points)."""
        keep = {}
        with torch.no_grad():
            if pe_pts is None:
                out = mod._fuse_native(enc_in.detach(), w, None, 0, keep=keep)
            else:
                width = 3 * (2 * n_freqs + 1)
                out = torch.empty(enc_in.shape[:-1] + (256 + width,), device=enc_in.device, dtype=torch.float32)
                mod._fuse_native(enc_in.detach(), w, out, 0, keep=keep)
                pos_encoding(pe_pts.detach(), n_freqs, out=out.reshape(-1, 256 + width), col_off=256)
        ctx.mod, ctx.w, ctx.in_shape, ctx.mask_col_const = mod, w, enc_in.shape, bool(mask_col_const)
        ctx.b_off, ctx.slope, ctx.am_x, ctx.am_fwd = keep['b_off'], keep['slope'], keep['am_x'], keep['am']
        ctx.save_for_backward(keep['x'], keep['net'], keep['e'], keep['s1'], keep['t1'], keep['scale'], *params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x, net, e, s1, t1, scale = ctx.saved_tensors[:6]
        W0, _, W1, _, Ws, Wsc1, _, Wsc2, _, Wsh1, _, Wsh2, _ = ctx.saved_tensors[6:]
        w, b_off, slope = ctx.w, ctx.b_off, ctx.slope
        need = ctx.needs_input_grad[6:]
        need_x = ctx.needs_input_grad[1]
        g_rows = grad_out.reshape(-1, grad_out.shape[-1]).float()
        if g_rows.stride(-1) != 1 or (g_rows.shape[0] > 1 and g_rows.stride(0) < g_rows.shape[1]) or g_rows.data_ptr() % 4:
            g_rows = g_rows.contiguous()
        ld_g = g_rows.stride(0) if g_rows.shape[0] > 1 else g_rows.shape[1]
        g = g_rows[:, :256]                                   # (a view when the node also produced the positional-encoding columns)
        dec = x[:, b_off:]
        mm = lambda a, b_: a.t() @ b_
        gp = [None] * 13
        if os.environ.get("E3DGE_FUSE_BWD", "hip") == "hip":
            # round 5: the data-gradient chain as nine e3dge_ws_linear launches on the transposed weight images; the thirteen parameter
            # gradients (reductions over the points: library GEMMs with K = number of points) only when a parameter wants one
            dz1, dz2, de, dnet, dx, am_b = ctx.mod._fuse_bwd_native(g, x, net, s1, t1, scale, ctx.am_x, w, b_off, slope, need_x, ld_g=ld_g,
                                                                    mask_col=not ctx.mask_col_const)
            if any(need):
                # the seven weight gradients: e3dge_wgrad (split-f16 MFMA, split over the points, fixed-order fold; E3DGE_WGRAD=library = matmul),
                # the relu of a layer's input folded into the operand load
                from .wgrad import amax_of, wgrad
                d_shift = g if w == 1.0 else w * g                # (a view of the incoming gradient rows when w = 1: e3dge_wgrad takes any row pitch)
                d_scale = d_shift * dec
                # amax buffers the two chains already hold: the forward's (x, net, e, s1, t1), the backward's (g, dz1, dz2, de, dnet); d shift = w g
                af = ctx.am_fwd
                am = {id(x): af[0], id(net): af[1], id(e): af[2], id(s1): af[3], id(t1): af[4],
                      id(dz1): am_b[1], id(dz2): am_b[2], id(de): am_b[3], id(dnet): am_b[4], id(d_shift): am_b[0] * abs(w)}

                gap = 256 if (b_off == 257 and x.shape[1] == 513) else None      # the visibility-mask column of the 513-wide input

                def wg(iw, ib, a_, b_, relu=False, gap_col=None):
                    """weight gradient iw and bias gradient ib (None: the layer has none) of one layer: ONE pass over the rows when both are wanted"""
                    want_b = ib is not None and need[ib]
                    if need[iw]:
                        for t_ in (a_, b_):
                            if id(t_) not in am:
                                am[id(t_)] = amax_of(t_)
                        r = wgrad(a_, b_, relu_b=relu, amax_a=am[id(a_)], amax_b=am[id(b_)], colsum=want_b, gap_col=gap_col)
                        if want_b:
                            gp[iw], gp[ib] = r
                        else:
                            gp[iw] = r
                    elif want_b:
                        gp[ib] = a_.sum(0)
                wg(7, 8, d_scale, s1)
                wg(11, 12, d_shift, t1)
                wg(5, 6, dz1, e)
                wg(9, 10, dz2, e)
                wg(2, 3, de, net, True)
                wg(4, None, de, x, gap_col=gap)
                wg(0, 1, dnet, x, True, gap_col=gap)
            return (None, dx.reshape(ctx.in_shape) if dx is not None else None, None, None, None, None, *gp)
        d_scale, d_shift = (w * g) * dec, w * g                   # out = dec + w (dec scale + shift)
        # scale = W2 lrelu(W1 e + b1) + b2 ; shift likewise
        dz1 = (d_scale @ Wsc2) * torch.where(s1 > 0, 1.0, slope)
        dz2 = (d_shift @ Wsh2) * torch.where(t1 > 0, 1.0, slope)
        if need[7]: gp[7] = mm(d_scale, s1)
        if need[8]: gp[8] = d_scale.sum(0)
        if need[11]: gp[11] = mm(d_shift, t1)
        if need[12]: gp[12] = d_shift.sum(0)
        if need[5]: gp[5] = mm(dz1, e)
        if need[6]: gp[6] = dz1.sum(0)
        if need[9]: gp[9] = mm(dz2, e)
        if need[10]: gp[10] = dz2.sum(0)
        de = dz1 @ Wsc1 + dz2 @ Wsh1
        # e = shortcut(x) + fc_1(relu(net)) + b1 ; net = fc_0(relu(x)) + b0
        dnet = (de @ W1) * (net > 0)
        if need[2]: gp[2] = mm(de, torch.relu(net))
        if need[3]: gp[3] = de.sum(0)
        if need[4]: gp[4] = mm(de, x)
        if need[0]: gp[0] = mm(dnet, torch.relu(x))
        if need[1]: gp[1] = dnet.sum(0)
        dx = None
        if need_x:
            dx = de @ Ws + (dnet @ W0) * (x > 0)
            dx[:, b_off:] += g * (1.0 + w * scale)
            dx = dx.reshape(ctx.in_shape)
        return (None, dx, None, None, None, None, *gp)


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
    if torch.is_grad_enabled() and (maps['ref'].requires_grad or maps['que'].requires_grad or pts.requires_grad or
                                    any(p_.requires_grad for p_ in fuse.parameters())):
        # training form (stage 2 trains the hourglass filters and Fuse_sft_MLP through this, e3dge_full_runner.py:185-317): the
        # same values assembled from differentiable pieces -- gathers with the HIP backward, Fuse_sft_MLP as the _FuseFn node (native
        # forward, written-out backward) or, when that does not apply, as torch modules
        vis_rows = None
        if add_mask:
            with torch.no_grad():
                surf = local_data_batch['xyz'].detach().reshape(B, 3, H * W).permute(0, 2, 1)
                _, vis_, _ = query_feature_map(surf, local_data_batch['ref_calibs'])
                vis_rows = vis_.reshape(B, H * W, 1).expand(B, H * W, S).reshape(B, N)
        if (not pts.requires_grad and isinstance(fuse, Fuse_sft_MLP) and fuse_autograd_backend() == "hip" and C == 256
                and os.environ.get("E3DGE_LOCAL_FEATS_NODES", "fused") == "fused"):
            # round 6: two nodes instead of eight -- the row buffers are written and read in place (E3DGE_LOCAL_FEATS_NODES=composed: as before)
            enc_in, in_img = _EncInFn.apply(pts, local_data_batch['que_calibs'], local_data_batch['ref_calibs'], vis_rows, maps['que'], maps['ref'])
            if fuse._native_ok(enc_in) and fuse._wants_grad(enc_in):
                feats = _FuseFn.apply(fuse, enc_in, 1.0, pts, n_freqs, True, *fuse._param_list())
                return feats.reshape(B, H, W, S, feats.shape[-1]), in_img.reshape(B, H, W, S, 1)
        a, _, _ = query_feature_map(pts, local_data_batch['que_calibs'], maps['que'])
        dec, in_img, _ = query_feature_map(pts, local_data_batch['ref_calibs'], maps['ref'])
        cols = [a]
        if add_mask:
            cols.append(vis_rows.reshape(B, N, 1))
        cols.append(dec)
        enc_in = torch.cat(cols, -1)
        fused = fuse.fuse(enc_in, enc_in[..., enc_in.shape[-1] - dec.shape[-1]:])      # (dec as the view of enc_in that it is)
        if pts.requires_grad:
            # the 45 positional-encoding columns carry a gradient to the points too (PosEncoding.forward, misc_utils.py:148-185):
            # plain differentiable torch ops here -- the kernel has no backward, and silently dropping it was an advisor finding
            enc_p = torch.cat([pts] + [fn(pts * float(2 ** k)) for k in range(n_freqs) for fn in (torch.sin, torch.cos)], -1)
        else:
            with torch.no_grad():
                enc_p = pos_encoding(pts.detach(), n_freqs)
        feats = torch.cat([fused, enc_p], -1)
        return feats.reshape(B, H, W, S, feats.shape[-1]), in_img.reshape(B, H, W, S, 1)
    enc_in = torch.empty((B, N, n_enc + C), device=pts.device, dtype=torch.float32)       # [2D-aligned | vis mask | 3D-projected]
    query_feature_map(pts, local_data_batch['que_calibs'], maps['que'], out=enc_in, col_off=0)
    dec, in_img, _ = query_feature_map(pts, local_data_batch['ref_calibs'], maps['ref'], out=enc_in, col_off=n_enc)
    if add_mask:   # is the query view's surface point inside the reference image? one value per ray, shared by its samples
        surf = local_data_batch['xyz'].reshape(B, 3, H * W).permute(0, 2, 1)
        _, vis, _ = query_feature_map(surf, local_data_batch['ref_calibs'])
        enc_in[..., C] = vis.reshape(B, H * W, 1).expand(B, H * W, S).reshape(B, N)
    width = 3 * (2 * n_freqs + 1)
    feats = torch.empty((B, N, C + width), device=pts.device, dtype=torch.float32)
    fuse.fuse(enc_in, dec, out=feats, out_off=0)
    pos_encoding(pts, n_freqs, out=feats.reshape(B * N, C + width), col_off=C)
    return feats.reshape(B, H, W, S, C + width), in_img.reshape(B, H, W, S, 1)


def tex_modulations_from_maps(local_head, renderer, cam_poses, focal, near, far, local_data_batch, lazy=False):
    """(alpha, beta), each (B,H,W,S,256), for the renderer's second pass.  `lazy=True` (VolumeFeatureRenderer.forward only, without an
    autograd graph): a private _LazyTex instead -- the renderer then decides between the fused head + FiLM launch on the record path
    and materialising (alpha, beta); external callers always get the documented tuple."""
    feats, in_img = local_features_from_maps(local_data_batch)
    local_data_batch['in_img_mask'] = in_img
    head = local_head.local_feat_to_tex_modulations_linear
    if lazy and not torch.is_grad_enabled():
        from .volume_renderer import _LazyTex
        return _LazyTex(head, feats)
    return head.tex_modulations(feats)
