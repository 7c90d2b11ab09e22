"""GPU: the packed decoder pipeline (csrc/decoder2.hip, e3dge_dec2_forward) -- Decoder.forward of
project/models/stylesdf_model.py:741-797 as one native call on split-f16 packed activations.

What is pinned here: (i) the packed <-> fp32 conversion, (ii) every intermediate activation against the planar (round-2)
kernels on the same GPU, (iii) the image against the CPU oracle (fp32 and float64) on shapes the oracle finishes in seconds,
incl. batch > 1, per-sample noise, tiles that overhang a small image, (iv) independence of the input magnitude (the operand
scale comes from an a-priori bound).  The reference goldens (decoder_256, generator_256, c3_eval_1024, generator_z_base) reach
this path through Decoder.forward in test_gpu_decoder.py / test_c3_eval.py.

Tolerance: image 1e-4 absolute on images of magnitude ~3, intermediates 2e-5 of the tensor's maximum."""
import os

import numpy as np
import pytest
import torch

from conftest import full_state_dict, maxerr, record
from oracle import decoder_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import _lib
from e3dge_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IMG_ATOL = 1e-4


def _planar(dec, feats, wd, noise):
    os.environ["E3DGE_DECODER"] = "planar"
    try:
        return dec(feats, [wd], input_is_latent=True, noise=noise, randomize_noise=False)[0]
    finally:
        os.environ.pop("E3DGE_DECODER", None)


def test_pack_unpack_roundtrip():
    lib = _lib.load()
    torch.manual_seed(0)
    for (B, C, R, mag) in ((1, 16, 8, 1.0), (2, 64, 33, 1e-4), (1, 32, 64, 3e4), (3, 8, 5, 1.0)):
        x = (torch.randn(B, C, R, R, device=DEV) * mag).contiguous()
        x[0, 0, 0, 0] = 0.0
        am = torch.zeros(_lib.AMAX_FLOATS, device=DEV)
        pk = torch.zeros(lib.e3dge_dec2_act_words(B, C, R), device=DEV, dtype=torch.int32)
        meta = torch.zeros(1, device=DEV, dtype=torch.int32)
        y = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.e3dge_amax(am.data_ptr(), x.data_ptr(), x.numel(), st), "amax")
        _lib.check(lib.e3dge_dec2_pack(pk.data_ptr(), meta.data_ptr(), x.data_ptr(), am.data_ptr(), B, C, R, st), "pack")
        _lib.check(lib.e3dge_dec2_unpack(y.data_ptr(), pk.data_ptr(), meta.data_ptr(), B, C, R, st), "unpack")
        rel = float((y - x).abs().max() / x.abs().max())
        record("dec2_pack_roundtrip", B=B, C=C, R=R, mag=mag, rel=rel)
        assert rel <= 2.0 ** -20
        # the one-entry border stays zero
        v = pk.view(B, C // 8, 2, R + 2, R + 2, 4)
        assert int(v[:, :, :, 0].abs().max()) == 0 and int(v[:, :, :, :, 0].abs().max()) == 0
        assert int(v[:, :, :, R + 1].abs().max()) == 0 and int(v[:, :, :, :, R + 1].abs().max()) == 0


@pytest.fixture(scope="module")
def gen256():
    g, sd = full_state_dict(size=256, cm=1)
    return g.to(DEV).eval(), sd


UPBLUR_FORMS = {                    # name -> (E3DGE_DEC2_UPBLUR, E3DGE_DEC2_UPBLUR_GEN, E3DGE_DEC2_UPBLUR_SHAPE)
    "auto": ("1", None, None),      # second generation, tile shape picked per level
    "gen2_8w_14x30": ("1", None, "0"), "gen2_8w_6x30": ("1", None, "1"), "gen2_4w_6x30": ("1", None, "2"), "gen2_4w_14x14": ("1", None, "3"),
    "gen2_8w_14x14": ("1", None, "4"),
    "gen1": ("1", "1", None),       # first generation (4x4 FIR from the LDS patch); also what a non-separable kernel takes
    "two_kernels": ("0", None, None),
}


@pytest.mark.parametrize("form", list(UPBLUR_FORMS))
def test_every_stage_against_the_planar_kernels(gen256, form, monkeypatch):
    """Up-sampling layer forms: transposed conv + blur in one kernel (T / H stay in LDS) in both generations and every tile
    shape of the second, and the two-kernel form with T in HBM."""
    upblur, gen, shape = UPBLUR_FORMS[form]
    monkeypatch.setenv("E3DGE_DEC2_UPBLUR", upblur)
    if gen is not None:
        monkeypatch.setenv("E3DGE_DEC2_UPBLUR_GEN", gen)
    if shape is not None:
        monkeypatch.setenv("E3DGE_DEC2_UPBLUR_SHAPE", shape)
    g, sd = gen256
    dec = g.decoder
    _, wd = syn.synthetic_inputs(1, seed=1, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = (0.5 * torch.randn(1, 256, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(3))).contiguous()
    noise = [getattr(dec.noises, f"noise_{i}") for i in range(dec.num_layers)]
    os.environ["E3DGE_DEC2_FUSE_RGB"] = "0"          # keep the last activation (the default build folds ToRGB into the last conv)
    with torch.no_grad():
        os.environ["E3DGE_DECODER"] = "planar"
        try:
            am = torch.zeros((dec.num_layers + 1, _lib.AMAX_FLOATS), device=DEV)
            mods = dec._all_modulations(wd)
            ref = [dec.conv1(feats, wd[:, 0], noise=noise[0], out_amax=am[1], pre=mods[0])]
            i, j = 1, 2
            for u in range(len(dec.to_rgbs)):
                ref.append(dec.convs[2 * u](ref[-1], wd[:, i], noise=noise[2 * u + 1], in_amax=am[i], out_amax=am[i + 1], pre=mods[j]))
                ref.append(dec.convs[2 * u + 1](ref[-1], wd[:, i + 1], noise=noise[2 * u + 2], in_amax=am[i + 1], out_amax=am[i + 2], pre=mods[j + 1]))
                i += 2
                j += 3
        finally:
            os.environ.pop("E3DGE_DECODER", None)
        ref_img = _planar(dec, feats, wd, noise)
        ms = []
        img = dec._forward_packed(feats, wd, noise, kernel_ms=ms)
        names = dec.dec2_launch_names()
        assert len(ms) == _lib.load().e3dge_dec2_num_launches(len(dec.to_rgbs)) == len(names)
        assert all((t == 0) == (upblur == "1" and n.endswith(".blur")) for n, t in zip(names, ms)), list(zip(names, ms))
        errs = {}
        for k, r in enumerate(ref):
            got = dec.dec2_unpack(1 + k, feats.shape)
            errs[f"act{1 + k}"] = maxerr(got, r) / float(r.abs().max())
        errs["img"] = maxerr(img, ref_img)
        os.environ.pop("E3DGE_DEC2_FUSE_RGB")
        img_fused = dec._forward_packed(feats, wd, noise)
        errs["img_fused_rgb"] = maxerr(img_fused, ref_img)
    record(f"dec2_stages_256_{form}", **errs)
    for k, v in errs.items():
        assert v <= (IMG_ATOL if k.startswith("img") else 2e-5), (k, v)
    # the default Decoder.forward takes the packed path and returns the same tensor values
    with torch.no_grad():
        img2, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
    assert torch.equal(img2, img_fused)


@pytest.mark.parametrize("B,res,size,per_sample_noise", [(2, 16, 128, True), (1, 8, 32, False), (3, 16, 64, False),
                                                         (2, 128, 512, True)])      # 256^2 level: 64 channels, NOT the last -> conv + store + ToRGB
def test_image_against_the_oracle_small_shapes(B, res, size, per_sample_noise):
    g, sd = full_state_dict(size=size, cm=1, res=res)
    g = g.to(DEV).eval()
    dec = g.decoder
    _, wd = syn.synthetic_inputs(B, seed=5, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    gen = torch.Generator(DEV).manual_seed(11)
    feats = (0.7 * torch.randn(B, 256, res, res, device=DEV, generator=gen)).contiguous()
    noise = []
    for i in range(dec.num_layers):
        r = getattr(dec.noises, f"noise_{i}").shape[-1]
        noise.append(torch.randn(B if per_sample_noise else 1, 1, r, r, device=DEV, generator=gen))
    with torch.no_grad():
        img, _ = dec(feats, [wd], input_is_latent=True, noise=noise, randomize_noise=False)
        planar = _planar(dec, feats, wd, noise)
    c = lambda t: t.detach().cpu()
    o32 = decoder_ref.decoder_forward(sd, c(feats), c(wd), noises=[c(n) for n in noise])
    o64 = decoder_ref.decoder_forward(sd, c(feats), c(wd), noises=[c(n) for n in noise], dtype=torch.float64)
    e = dict(vs_f32=maxerr(img, o32), vs_f64=maxerr(img, o64), f32_vs_f64=maxerr(o32, o64), planar_vs_f64=maxerr(planar, o64),
             img_max=float(o64.abs().max()))
    record(f"dec2_oracle_B{B}_res{res}_size{size}", **e)
    assert tuple(img.shape) == (B, 3, size, size)
    assert e['vs_f32'] <= IMG_ATOL and e['vs_f64'] <= max(IMG_ATOL, 3 * e['f32_vs_f64'])


def test_input_magnitude_does_not_matter(gen256):
    """The operand scale of every packed tensor comes from a bound; scaling the feature map by 2^k scales conv1's output
    only through the noise / bias terms -- compare against the planar path at each magnitude instead of assuming linearity."""
    g, sd = gen256
    dec = g.decoder
    _, wd = syn.synthetic_inputs(1, seed=2, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    base = torch.randn(1, 256, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(4))
    noise = [getattr(dec.noises, f"noise_{i}") for i in range(dec.num_layers)]
    out = {}
    for mag in (1e-5, 1.0, 1e4):
        feats = (base * mag).contiguous()
        with torch.no_grad():
            img, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
            ref = _planar(dec, feats, wd, noise)
        out[str(mag)] = maxerr(img, ref) / float(ref.abs().max())
        assert torch.isfinite(img).all()
    record("dec2_magnitude", **out)
    assert max(out.values()) <= 2e-5


def test_fresh_caller_noise_is_measured_every_call(gen256):
    """Advisor finding (round 3): max|noise| was cached by (data_ptr, version); fresh caller tensors of one size are recycled by
    the caching allocator with the same version, so a LARGE noise could inherit the maximum of an earlier small one and the
    operand-scale bound (|noise_w| amax_noise + ...) came out too small -- inf / garbage without an error.  Ad-hoc noise is
    now measured on every call; only the module's own `noises.noise_i` buffers are cached."""
    g, sd = gen256
    dec = g.decoder
    _, wd = syn.synthetic_inputs(1, seed=2, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = torch.randn(1, 256, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(5)).contiguous()
    shapes = [tuple(getattr(dec.noises, f"noise_{i}").shape) for i in range(dec.num_layers)]
    gen = torch.Generator(DEV).manual_seed(6)
    worst, ptrs = 0.0, []
    with torch.no_grad():
        for mag in (1.0, 3e3, 1.0, 1e4):                    # small, LARGE on the recycled blocks, small, larger
            noise = [(torch.randn(s, device=DEV, generator=gen) * mag).contiguous() for s in shapes]
            ptrs.append(noise[-1].data_ptr())
            img, _ = dec(feats, [wd], input_is_latent=True, noise=noise, randomize_noise=False)
            ref = _planar(dec, feats, wd, noise)
            assert torch.isfinite(img).all()
            worst = max(worst, maxerr(img, ref) / float(ref.abs().max()))
            del noise, img, ref
    record("dec2_fresh_noise", rel=worst, blocks_recycled=len(set(ptrs)) < len(ptrs))
    assert worst <= 2e-5


def test_falls_back_when_a_graph_is_needed():
    """A latent that requires grad must not take the packed (graph-less) path: the gradient w.r.t. the decoder latent exists
    and matches the library path (fused kernels silently dropped dL/d(style))."""
    _, wd = syn.synthetic_inputs(1, seed=2, device=DEV)
    feats = 0.5 * torch.randn(1, 256, 16, 16, device=DEV)
    gs, _ = full_state_dict(size=64, cm=1, res=16)          # a small decoder keeps the library path cheap
    ds = gs.decoder.to(DEV).eval()
    for p in ds.parameters():
        p.requires_grad_(False)
    wl = wd[:, :ds.n_latent].detach().clone().requires_grad_(True)
    img, _ = ds(feats, [wl], input_is_latent=True, randomize_noise=False)
    assert img.requires_grad
    (img ** 2).mean().backward()
    assert wl.grad is not None and float(wl.grad.abs().max()) > 0
    os.environ["E3DGE_MODCONV"] = "library"
    try:
        wl2 = wl.detach().clone().requires_grad_(True)
        img2, _ = ds(feats, [wl2], input_is_latent=True, randomize_noise=False)
        (img2 ** 2).mean().backward()
    finally:
        os.environ.pop("E3DGE_MODCONV", None)
    rel = float((wl.grad - wl2.grad).abs().max() / wl2.grad.abs().max())
    record("dec2_latent_grad_fallback", rel=rel)
    assert rel <= 1e-4


def test_packed_forward_inside_an_autograd_graph(gen256, monkeypatch):
    """E3DGE_DECODER_AUTOGRAD=packed: a forward that needs a graph (train_ae.py: the renderer's features and the predicted latent
    require grad, trainer.py:881-897) still runs the packed pipeline; its backward differentiates the recomputed library path.
    The image equals the no-grad packed image bit for bit; gradients w.r.t. features, latent and a parameter equal the library
    path's for the same upstream gradient (same backward graph), and match float64 autograd of the oracle on a small decoder."""
    g, sd = gen256
    dec = g.decoder
    for p_ in dec.parameters():
        p_.requires_grad_(False)
    _, wd = syn.synthetic_inputs(1, seed=2, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = torch.randn(1, 256, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(8)).contiguous()
    wgt = torch.randn(1, 3, 256, 256, device=DEV, generator=torch.Generator(DEV).manual_seed(9))
    with torch.no_grad():
        ref_img, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
    dec.conv1.activate.bias.requires_grad_(True)

    def run(backend):
        monkeypatch.setenv("E3DGE_DECODER_AUTOGRAD", backend)
        f = feats.clone().requires_grad_(True)
        l = wd.clone().requires_grad_(True)
        dec.conv1.activate.bias.grad = None
        img, _ = dec(f, [l], input_is_latent=True, randomize_noise=False)
        (img * wgt).sum().backward()
        return img.detach(), f.grad, l.grad, dec.conv1.activate.bias.grad.clone()
    try:
        img_p, gf_p, gl_p, gb_p = run("packed")
        img_l, gf_l, gl_l, gb_l = run("library")
    finally:
        dec.conv1.activate.bias.requires_grad_(False)
    assert torch.equal(img_p, ref_img)                              # the packed forward, not the library one
    assert maxerr(img_l, ref_img) <= IMG_ATOL
    e = dict(d_features=maxerr(gf_p, gf_l) / float(gf_l.abs().max()), d_latent=maxerr(gl_p, gl_l) / float(gl_l.abs().max()),
             d_bias=maxerr(gb_p, gb_l) / float(gb_l.abs().max()))
    record("dec2_autograd_packed_vs_library", **e)
    assert max(e.values()) <= 1e-6, e
    # random noise: the backward must see the noise the forward drew (the image depends on it)
    monkeypatch.setenv("E3DGE_DECODER_AUTOGRAD", "packed")
    f = feats.clone().requires_grad_(True)
    img_r, _ = dec(f, [wd], input_is_latent=True, randomize_noise=True)
    img_r.square().mean().backward()
    assert torch.isfinite(f.grad).all() and float(f.grad.abs().max()) > 0


def test_a_decoder_that_ran_the_packed_path_is_still_deep_copyable_and_the_copy_runs(gen256):
    """Runners deep-copy generators (EMA / surface copies): the native plan (ctypes pointers) lives outside the module."""
    import copy
    g, sd = gen256
    dec = g.decoder
    _, wd = syn.synthetic_inputs(1, seed=1, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = 0.5 * torch.randn(1, 256, 64, 64, device=DEV)
    with torch.no_grad():
        a, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
        dec2 = copy.deepcopy(dec)
        b, _ = dec2(feats, [wd], input_is_latent=True, randomize_noise=False)
    assert torch.equal(a, b)


def test_local_query_keeps_the_gradient_or_refuses():
    """ADVICE r2 / VERDICT r3 #8: a map (or points) that requires grad is differentiated (e3dge_local_query_bwd) -- or, for the
    outputs that carry no gradient here (projection, in-place slices), refused; never silently detached."""
    from e3dge_amd.local_query import query_feature_map
    pts = torch.rand(1, 16, 3, device=DEV)
    calib = torch.eye(4, device=DEV)[None, :3].contiguous()
    fmap = torch.randn(1, 8, 4, 4, device=DEV, requires_grad=True)
    f, m, _ = query_feature_map(pts, calib, fmap)
    assert f.requires_grad and not m.requires_grad
    f.sum().backward()
    assert fmap.grad is not None and torch.isfinite(fmap.grad).all()
    with pytest.raises(NotImplementedError):
        query_feature_map(pts, calib, fmap, want_proj=True)
    with torch.no_grad():
        f, m, _ = query_feature_map(pts, calib, fmap)
    assert tuple(f.shape) == (1, 16, 8)


def test_stale_render_event_is_dropped():
    """a grad-mode render that never consumed its completion event must not order (or break deepcopy of) a later call."""
    import copy
    from test_gpu_renderer import make_renderer
    from e3dge_amd.camera_utils import generate_camera_params
    sd = full_state_dict(res=8, n_samples=18)[1]
    r = make_renderer(sd, 8, 18)
    wr, _ = syn.synthetic_inputs(1, seed=1, device=DEV)
    p, f, n, fa, _ = generate_camera_params(8, DEV, locations=torch.zeros(1, 2, device=DEV))
    s = wr.clone().requires_grad_(True)
    r(p, f, n, fa, styles=s, return_eikonal=True)                 # grad mode, no surface query: leaves an event behind
    with torch.no_grad():
        o = r(p, f, n, fa, styles=wr, return_eikonal=True, return_surface_eikonal=True)
    assert getattr(r, '_render_done', None) is None and torch.isfinite(o['surface_eikonal_term']).all()
    copy.deepcopy(r)


def _fuse_module(in_ch, seed):
    from e3dge_amd.local_query import Fuse_sft_MLP
    torch.manual_seed(seed)
    m = Fuse_sft_MLP(in_ch=in_ch, out_ch=256)
    with torch.no_grad():
        for p in m.parameters():                                   # (the reference zero-initialises fc_1.weight and the biases)
            p.copy_(torch.randn_like(p) * (0.3 if p.ndim == 1 else 1.5 / p.shape[1] ** 0.5))
    return m


@pytest.mark.parametrize("in_ch,n_pts,mag", [(257, 1000, 1.0), (257, 64, 300.0), (256, 333, 1e-3), (257, 4096, 1.0)])
def test_fuse_sft_mlp_native_against_float64(in_ch, n_pts, mag, monkeypatch):
    """Fuse_sft_MLP (sft.py:84-109, resnetfc.py:49-58) as nine e3dge_ws_linear launches vs the same module in float64 on the CPU;
    tolerance 3 x what the fp32 torch modules deviate (+ 1e-6 of the output's maximum), ragged row counts, both input widths,
    large / small magnitudes (the operand scale comes from the tensors' amax), writing into a column slice of a wider buffer."""
    import copy
    m = _fuse_module(in_ch, seed=in_ch + n_pts)
    m64 = copy.deepcopy(m).double()
    torch.manual_seed(1)
    enc_in = mag * torch.randn(2, n_pts, in_ch + 256)
    if in_ch == 257:
        enc_in[..., 256] = (torch.rand(2, n_pts) > 0.4).float()
    dec = enc_in[..., in_ch:]
    with torch.no_grad():
        ref = m64.fuse(enc_in.double(), dec.double(), w=0.7)
        r32 = m.fuse(enc_in, dec, w=0.7)
        mg = m.to(DEV)
        xg = enc_in.to(DEV)
        wide = torch.full((2, n_pts, 301), 7.0, device=DEV)
        got = mg.fuse(xg, xg[..., in_ch:], w=0.7, out=wide, out_off=0)
        monkeypatch.setenv("E3DGE_FUSE", "torch")
        lib = mg.fuse(xg, xg[..., in_ch:], w=0.7)
        monkeypatch.delenv("E3DGE_FUSE")
    scale = float(ref.abs().max())
    err, e32, elib = (float((t.cpu().double() - ref).abs().max()) for t in (got, r32, lib))
    record("fuse_sft_mlp", in_ch=in_ch, n=n_pts, mag=mag, err=err, fp32_cpu=e32, torch_gpu=elib, out_max=scale)
    assert err <= 3 * e32 + 1e-6 * scale, (err, e32, scale)
    assert float((wide[..., 256:] - 7.0).abs().max()) == 0.0                  # columns beyond the slice untouched
    with torch.enable_grad():                                                  # a graph is needed: the torch modules run
        xg2 = xg.clone().requires_grad_(True)
        y = mg.fuse(xg2, xg2[..., in_ch:], w=0.7)
        assert y.requires_grad


@pytest.mark.parametrize("in_ch,n_pts", [(257, 777), (256, 130)])
def test_fuse_sft_mlp_under_autograd_against_float64(in_ch, n_pts, monkeypatch):
    """Fuse_sft_MLP with a graph (stage-2 training, e3dge_full_runner.py:185-317): the forward is the nine weight-stationary launches
    (_FuseFn), the backward the written-out chain rule of sft.py:84-109 + resnetfc.py:49-58.  d(input) and all thirteen parameter
    gradients against float64 autograd of the same network on the CPU.  relu / leaky relu have kinks: a pre-activation within rounding
    of zero may sit on the other side in float64, and ONE such flip moves a few gradient entries by O(1) -- so the float64 reference
    takes its activation pattern from the forward under test (the derivative of the function that was actually computed), and the
    forward itself is held to float64 separately.  Tolerance 2e-5 of each gradient's maximum."""
    import copy
    import torch.nn.functional as F
    m = _fuse_module(in_ch, seed=11 + n_pts)
    m64 = copy.deepcopy(m).double()
    torch.manual_seed(5)
    enc_in = torch.randn(2, n_pts, in_ch + 256)
    if in_ch == 257:
        enc_in[..., 256] = (torch.rand(2, n_pts) > 0.4).float()
    gy = torch.randn(2, n_pts, 256)
    mg = m.to(DEV)
    # the forward under test, its intermediates, and the plain float64 forward
    keep = {}
    with torch.no_grad():
        y_nat = mg._fuse_native(enc_in.to(DEV), 0.7, None, 0, keep=keep)
        y64 = m64.fuse(enc_in.double(), enc_in.double()[..., in_ch:], w=0.7)
    assert float((y_nat.cpu().double() - y64).abs().max()) <= 2e-5 * float(y64.abs().max())
    m_net, m_s1, m_t1 = ((keep[k] > 0).cpu().reshape(2, n_pts, 256) for k in ("net", "s1", "t1"))
    # float64 autograd with that activation pattern
    x64 = enc_in.double().requires_grad_(True)
    e_ = m64.encode_enc
    net = e_.fc_0(x64 * (x64 > 0))
    e = e_.shortcut(x64) + e_.fc_1(net * m_net)
    lk = lambda z, msk: z * torch.where(msk, 1.0, 0.2).double()
    sc = m64.scale[2](lk(m64.scale[0](e), m_s1))
    sh = m64.shift[2](lk(m64.shift[0](e), m_t1))
    dec = x64[..., in_ch:]
    y_ref = dec + 0.7 * (dec * sc + sh)
    assert float((y_ref.detach() - y64).abs().max()) <= 1e-4 * float(y64.abs().max())    # (the patterns differ in at most a few entries)
    (y_ref * gy.double()).sum().backward()
    ref = [x64.grad] + [p.grad for p in m64.parameters()]
    # the node under test
    xg = enc_in.to(DEV).requires_grad_(True)
    y = mg.fuse(xg, xg[..., in_ch:], w=0.7)
    assert "_FuseFn" in type(y.grad_fn).__name__
    assert float((y.detach() - y_nat).abs().max()) == 0.0
    (y * gy.to(DEV)).sum().backward()
    got = [xg.grad] + [p.grad for p in mg.parameters()]
    labels = ["d_input"] + [n for n, _ in m.named_parameters()]
    assert labels[1:] == [
        "encode_enc.fc_0.weight", "encode_enc.fc_0.bias", "encode_enc.fc_1.weight", "encode_enc.fc_1.bias", "encode_enc.shortcut.weight",
        "scale.0.weight", "scale.0.bias", "scale.2.weight", "scale.2.bias", "shift.0.weight", "shift.0.bias", "shift.2.weight", "shift.2.bias"]
    worst = 0.0
    for lab, r, a in zip(labels, ref, got):
        assert a is not None and a.shape == r.shape, lab
        scale = float(r.abs().max())
        err = float((a.cpu().double() - r).abs().max())
        worst = max(worst, err / max(scale, 1e-30))
        assert err <= 2e-5 * scale, (lab, err, scale)
    record("fuse_sft_mlp_autograd", in_ch=in_ch, n=n_pts, worst_rel_err=worst)
    # E3DGE_FUSE_AUTOGRAD=torch and an `out` buffer under autograd take the module path
    monkeypatch.setenv("E3DGE_FUSE_AUTOGRAD", "torch")
    x2 = enc_in.to(DEV).requires_grad_(True)
    assert "_FuseFn" not in type(mg.fuse(x2, x2[..., in_ch:], w=0.7).grad_fn).__name__
    monkeypatch.delenv("E3DGE_FUSE_AUTOGRAD")
    wide = torch.zeros(2, n_pts, 301, device=DEV)
    y2 = mg.fuse(x2, x2[..., in_ch:], w=0.7, out=wide, out_off=0)
    assert y2.requires_grad and "_FuseFn" not in type(y2.grad_fn).__name__


def test_mid_level_torgb_in_the_epilogue_matches_the_stand_alone_launch(monkeypatch):
    """512^2 decoder from 128^2 features: the 256^2 level has 64 channels and is not the last, so its ToRGB rides in the convolution's
    epilogue (pkconv_s1_kernel, RGB = 2: activation stored AND reduced).  Against the same forward with the stand-alone ToRGB launch
    (E3DGE_DEC2_FUSE_RGB_MID=0): the two sum the 64 channels in different orders -- 1e-5 of the image's range; and the stored
    activation of that level is the same bits either way (the next level reads it)."""
    g, _ = full_state_dict(size=512, cm=1, res=128)
    dec = g.to(DEV).eval().decoder
    _, wd = syn.synthetic_inputs(1, seed=9, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = (0.7 * torch.randn(1, 256, 128, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(3))).contiguous()
    with torch.no_grad():
        img, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
        act = dec.dec2_unpack(3, feats.shape).clone()                 # output of the 256^2 level's stride-1 convolution
        monkeypatch.setenv("E3DGE_DEC2_FUSE_RGB_MID", "0")
        img0, _ = dec(feats, [wd], input_is_latent=True, randomize_noise=False)
        act0 = dec.dec2_unpack(3, feats.shape)
    assert tuple(act.shape) == (1, 64, 256, 256) and float((act - act0).abs().max()) == 0.0
    err, scale = float((img - img0).abs().max()), float(img0.abs().max())
    record("dec2_mid_torgb", err=err, img_max=scale)
    assert err <= 1e-5 * max(scale, 1.0)
