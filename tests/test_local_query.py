"""the per-point query of the local branch's feature maps -- projection + bilinear gather + in-image masks +
positional encoding (HIP) + Fuse_sft_MLP -- against tests/golden/localquery_8x24.npz, recorded from the reference's own
perspective / index / PosEncoding / Fuse_sft_MLP composed as que_render_given_ref does (oracle/gen_golden_localquery.py).

Stated fp32 tolerance: gathered features are O(3); the projection is evaluated in a different fp32 order than BLAS' baddbmm
(coordinates differ by ~1e-7, times the map's gradient ~ 30 per unit).  Bound: gathers 2e-5, positional encoding 2e-6,
fused 301-channel features 1e-4 (the reference itself is 2.3e-5 from float64), masks identical."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxerr, record
from oracle import local_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn

PREFIX = 'Fuse_sft_block.'


def fuse_state():
    from e3dge_amd.local_query import Fuse_sft_MLP
    m = Fuse_sft_MLP(257, 256)
    sd = {}
    for k, v in m.state_dict().items():
        t = syn.synthetic_tensor(PREFIX + k, v.shape)
        sd[k] = t / np.sqrt(v.shape[1]) if k.endswith('weight') else t
    m.load_state_dict(sd)
    return m, {PREFIX + k: v for k, v in sd.items()}


def maps_of(g):
    rs = np.random.RandomState(int(g['maps_seed']))
    shp = tuple(int(v) for v in g['map_shape'])
    ref_map = torch.from_numpy(rs.standard_normal(shp).astype(np.float32))
    que_map = torch.from_numpy(rs.standard_normal(shp).astype(np.float32))
    return ref_map, que_map


def test_oracle_reproduces_the_reference_local_features():
    g = load_golden("localquery_8x24")
    _, sd = fuse_state()
    ref_map, que_map = maps_of(g)
    T = torch.from_numpy
    with torch.no_grad():
        feats, mask = local_ref.local_features(sd, PREFIX, T(g['points']), T(g['xyz']), ref_map, que_map, T(g['ref_calibs']), T(g['que_calibs']))
        q = local_ref.query(T(g['points']).reshape(2, -1, 3).permute(0, 2, 1), T(g['ref_calibs']), ref_map)
    assert maxerr(feats[:, :, :, ::4], g['ref_feats_s4']) == 0
    assert np.array_equal(mask.reshape(2, -1).numpy(), g['ref_in_img'])
    assert maxerr(q['proj_xy'], g['ref_proj_xy']) == 0 and maxerr(q['depth'], g['ref_depth']) == 0


@pytest.mark.gpu
def test_local_query_kernels_and_pipeline_on_gpu():
    from e3dge_amd.local_query import local_features_from_maps, pos_encoding, query_feature_map
    dev = "cuda:0"
    g = load_golden("localquery_8x24")
    fuse, sd = fuse_state()
    fuse = fuse.to(dev).eval()
    ref_map, que_map = maps_of(g)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    pts5 = T('points')
    B, H, W, S, _ = pts5.shape
    pts = pts5.reshape(B, -1, 3)
    with torch.no_grad():
        f3, in_img, proj = query_feature_map(pts, T('ref_calibs'), ref_map.to(dev), want_proj=True)
        f2, _, _ = query_feature_map(pts, T('que_calibs'), que_map.to(dev))
        pe = pos_encoding(pts5)
        feats, mask = local_features_from_maps(dict(feature_maps=dict(ref=ref_map.to(dev), que=que_map.to(dev)), ref_calibs=T('ref_calibs'),
                                                    que_calibs=T('que_calibs'), points=pts5, xyz=T('xyz'), fuse_sft_block=fuse))
    e = dict(f3=maxerr(f3.reshape(B, H, W, S, -1)[:, :, :, ::12], g['ref_feature_3dprojection_s12']),
             f2=maxerr(f2.reshape(B, H, W, S, -1)[:, :, :, ::12], g['ref_feature_2dalign_s12'][..., :256]),
             proj_xy=maxerr(proj[..., :2].permute(0, 2, 1), g['ref_proj_xy']), depth=maxerr(proj[..., 2:3].permute(0, 2, 1), g['ref_depth']),
             pe=maxerr(pe[:, :, :, ::4], g['ref_pe_s4']), feats=maxerr(feats[:, :, :, ::4], g['ref_feats_s4']),
             feats_vs_f64=maxerr(feats[:, :, :, ::4], g['f64_feats_s4']),
             ref_vs_f64=float(np.abs(g['ref_feats_s4'] - g['f64_feats_s4']).max()))
    record("localquery_8x24", **e)
    # points whose projection sits within 1e-6 of the image border may flip the mask; none do on this fixture
    assert np.array_equal(in_img.cpu().numpy().astype(bool), g['ref_in_img']) and torch.equal(mask.reshape(B, -1), in_img)
    assert e['proj_xy'] <= 2e-6 and e['depth'] <= 1e-6 and e['pe'] <= 2e-6, e
    assert e['f3'] <= 2e-5 and e['f2'] <= 2e-5, e
    assert e['feats'] <= 1e-4 and e['feats_vs_f64'] <= 3 * e['ref_vs_f64'] + 2e-5, e
    assert tuple(feats.shape) == (B, H, W, S, 301)
    # edge cases: projection only, points far outside every image (zeros padding), empty point set
    far = torch.tensor([[[5.0, 5.0, 0.0], [0.0, 0.0, 0.0]]], device=dev)
    with torch.no_grad():
        fz, mz, _ = query_feature_map(far, T('ref_calibs')[:1], ref_map[:1].to(dev))
        none, m0, _ = query_feature_map(pts[:, :0], T('ref_calibs'), ref_map.to(dev))
        only_mask = query_feature_map(pts, T('ref_calibs'))[1]
    assert float(fz[0, 0].abs().max()) == 0.0 and float(mz[0, 0]) == 0.0 and none.shape == (B, 0, 256)
    assert torch.equal(only_mask, in_img)


@pytest.mark.gpu
def test_second_pass_from_feature_maps():
    """VolumeFeatureRenderer.forward with local_data_batch={'feature_maps': ...}: query kernels -> Fuse_sft_MLP -> fused
    texture head -> tex-FiLM render, against the oracle's render with the oracle's (alpha, beta)."""
    from conftest import full_state_dict
    from oracle import renderer_ref
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    dev = "cuda:0"
    g = load_golden("localquery_8x24")
    res, S = 8, 24
    fuse, fsd = fuse_state()
    fuse = fuse.to(dev).eval()
    _, sd = full_state_dict(res=res, n_samples=S)
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), out_im_res=res, mode='test')
    TP = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
    own = {}
    for k in r.state_dict():
        own[k] = 0.05 * syn.synthetic_tensor('renderer.' + k, r.state_dict()[k].shape) if 'netLocal' in k else sd['renderer.' + k.replace('network.netGlobal.', 'network.')]
    r.load_state_dict(own)
    r = r.to(dev)
    sd_all = dict(sd)
    sd_all.update({'renderer.' + k: v for k, v in own.items() if 'netLocal' in k})
    ref_map, que_map = maps_of(g)
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    wr, _ = syn.synthetic_inputs(2, seed=int(g['styles_seed']), device=dev)
    cam = (T('poses'), T('focal'), T('near'), T('far'))
    with torch.no_grad():
        p1 = r(*cam, styles=wr)
        out = r(*cam, styles=wr, local_data_batch=dict(feature_maps=dict(ref=ref_map.to(dev), que=que_map.to(dev)), ref_calibs=T('ref_calibs'),
                                                       que_calibs=T('que_calibs'), points=p1['points'], xyz=p1['xyz'], fuse_sft_block=fuse))
        c = lambda t: t.detach().cpu()
        feats, _ = local_ref.local_features(fsd, PREFIX, c(p1['points']), c(p1['xyz']), ref_map, que_map, c(T('ref_calibs')), c(T('que_calibs')))
        tex = renderer_ref.tex_modulations(sd_all, TP, feats)
        ref = renderer_ref.render(sd, *[c(t) for t in cam], c(wr), res=res, n_samples=S, tex=tex)
    e = dict(features=maxerr(out['features'], ref['features']), rgb=maxerr(out['gen_thumb_imgs'], ref['gen_thumb_imgs']),
             tex_effect=maxerr(ref['features'], c(p1['features'])))
    record("second_pass_from_feature_maps", **e)
    assert e['tex_effect'] > 1e-2 and e['features'] <= 1e-4 and e['rgb'] <= 5e-6, e


@pytest.mark.gpu
@pytest.mark.parametrize("C,h,w,N", [(8, 5, 7, 300), (256, 32, 32, 2000)])
def test_gather_backward_against_float64_grid_sample(C, h, w, N):
    """e3dge_local_query_bwd (the reference's grid_sample_gradfix backward, project/models/op/grid_sample_gradfix.py:52-89, behind
    HGPIFuNetGAN.query, vendor/pifu/lib/model/HGPIFuGANNet.py:85-151): d feature map and d points against float64 autograd of
    perspective + F.grid_sample(bilinear, zeros padding, align_corners=False) on the same inputs, incl. points that project
    outside the map (zero padding on some corners) and far outside (no gradient at all).
    Tolerance: 2e-5 of the gradient's maximum (fp32 atomics in arbitrary order; measured ~1e-6)."""
    import torch.nn.functional as F
    from e3dge_amd.local_query import query_feature_map
    DEV = "cuda:0"
    gen = torch.Generator("cpu").manual_seed(5)
    B = 2
    pts = (torch.rand(B, N, 3, generator=gen) - 0.5) * 0.5
    pts[:, :N // 8] *= 6.0                                          # some project outside the image plane
    calib = torch.tensor([[[2.2, 0.1, 0.0, 0.03], [0.05, 2.1, 0.1, -0.02], [0.0, 0.1, 1.0, 2.0]],
                          [[1.9, -0.2, 0.1, 0.0], [0.1, 2.3, 0.0, 0.05], [0.1, 0.0, 1.0, 1.7]]])
    fmap = torch.randn(B, C, h, w, generator=gen)
    g_out = torch.randn(B, N, C, generator=gen)

    def ref(p, fm):
        homo = torch.einsum('bij,bnj->bni', calib.double()[:, :, :3], p) + calib.double()[:, None, :, 3]
        hz0 = float(homo[0, 0, 2])
        z = -homo[..., 2] if hz0 < 0 else homo[..., 2]
        xy = homo[..., :2] / z[..., None]
        uv = torch.stack([xy[..., 0], -xy[..., 1]], -1)
        return F.grid_sample(fm, uv[:, :, None, :], mode='bilinear', padding_mode='zeros', align_corners=False)[..., 0].permute(0, 2, 1)
    p64 = pts.double().requires_grad_(True)
    f64 = fmap.double().requires_grad_(True)
    out64 = ref(p64, f64)
    (out64 * g_out.double()).sum().backward()
    p32 = pts.to(DEV).requires_grad_(True)
    f32 = fmap.to(DEV).requires_grad_(True)
    out, mask, _ = query_feature_map(p32, calib.to(DEV), f32)
    (out * g_out.to(DEV)).sum().backward()
    e = dict(fwd=float((out.detach().cpu().double() - out64.detach()).abs().max()),
             d_fmap=float((f32.grad.cpu().double() - f64.grad).abs().max() / f64.grad.abs().max()),
             d_pts=float((p32.grad.cpu().double() - p64.grad).abs().max() / p64.grad.abs().max()))
    from conftest import record
    record("local_query_backward", C=C, h=h, w=w, N=N, **e)
    assert e['fwd'] <= 2e-5 and e['d_fmap'] <= 2e-5 and e['d_pts'] <= 1e-4, e
    # round 6: the same scatter with the points walked in pixel order (e3dge_local_query_bwd_sorted: counting sort by corner pixel, then
    # the same kernel on `order`) -- d map against float64 at the same bound, d points bit-identical (per-point arithmetic, order-free)
    from e3dge_amd import _lib
    lib = _lib.load()
    gd = g_out.to(DEV).contiguous()
    fm = fmap.to(DEV).permute(0, 2, 3, 1).contiguous()
    pd, cd = pts.to(DEV).contiguous(), calib.to(DEV).contiguous()
    res = {}
    for name in ("plain", "sorted"):
        d_fm = torch.zeros(B, h, w, C, device=DEV)
        d_p = torch.empty(B, N, 3, device=DEV)
        if name == "plain":
            rc = lib.e3dge_local_query_bwd(_lib.ptr(d_fm), _lib.ptr(d_p), _lib.ptr(gd), C, 0, _lib.ptr(pd), _lib.ptr(cd), _lib.ptr(fm), B, N, C, h, w, None)
        else:
            n_ws = lib.e3dge_local_query_sort_ws_ints(B, N, h, w)
            ws = torch.empty(n_ws, device=DEV, dtype=torch.int32)
            rc = lib.e3dge_local_query_bwd_sorted(_lib.ptr(d_fm), _lib.ptr(d_p), _lib.ptr(gd), C, 0, _lib.ptr(pd), _lib.ptr(cd), _lib.ptr(fm), B, N, C, h, w,
                                                  _lib.ptr(ws), n_ws, None)
            order = ws[B * N:2 * B * N].long()
            assert torch.equal(torch.sort(order).values, torch.arange(B * N, device=DEV)), "order is not a permutation"
        assert rc == 0
        torch.cuda.synchronize()
        res[name] = (d_fm.permute(0, 3, 1, 2).cpu().double(), d_p.clone())
    e2 = float((res["sorted"][0] - f64.grad).abs().max() / f64.grad.abs().max())
    record("local_query_backward_sorted", C=C, h=h, w=w, N=N, d_fmap=e2)
    assert e2 <= 2e-5 and torch.equal(res["sorted"][1], res["plain"][1]), e2


@pytest.mark.gpu
def test_training_form_of_the_local_features_matches_the_inference_form():
    """With a feature map that requires grad, local_features_from_maps assembles the same (B,H,W,S,301) features from
    differentiable pieces (gathers with the HIP backward + Fuse_sft_MLP as torch modules): values equal the inference form
    (native Fuse_sft_MLP) to 1e-4, and the gradient reaches both maps and the fuse block's parameters."""
    from e3dge_amd.local_query import Fuse_sft_MLP, local_features_from_maps
    DEV = "cuda:0"
    torch.manual_seed(3)
    B, H, S, C = 1, 8, 6, 256
    fuse = Fuse_sft_MLP(C + 1, C).to(DEV)
    for p_ in fuse.parameters():
        torch.nn.init.normal_(p_, std=0.05)
    maps = {'ref': torch.randn(B, C, 16, 16, device=DEV), 'que': torch.randn(B, C, 16, 16, device=DEV)}
    calib = torch.tensor([[[2.0, 0.0, 0.0, 0.0], [0.0, 2.0, 0.0, 0.0], [0.0, 0.0, 1.0, 2.0]]], device=DEV)
    batch = dict(feature_maps=maps, ref_calibs=calib, que_calibs=calib.clone(), points=(torch.rand(B, H, H, S, 3, device=DEV) - 0.5),
                 xyz=(torch.rand(B, 3, H, H, device=DEV) - 0.5), fuse_sft_block=fuse)
    with torch.no_grad():
        want, _ = local_features_from_maps(batch)
    maps['ref'].requires_grad_(True); maps['que'].requires_grad_(True)
    got, _ = local_features_from_maps(batch)
    assert got.requires_grad and float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    got.square().mean().backward()
    assert all(m.grad is not None and torch.isfinite(m.grad).all() and float(m.grad.abs().max()) > 0 for m in maps.values())
    assert all(p_.grad is not None for p_ in fuse.parameters())


@pytest.mark.gpu
def test_two_node_training_form_equals_the_composed_one(monkeypatch):
    """Round 6: the training form as two nodes (_EncInFn: both gathers into the row buffer, their backward on column blocks of the incoming
    gradient in place; _FuseFn with the positional-encoding columns in its own output, the first 256 gradient columns read in place) against
    the composed form (E3DGE_LOCAL_FEATS_NODES=composed: gathers, cat, _FuseFn, cat): identical values, the same gradients for both maps
    and all thirteen parameters to rounding -- the loss weights the 45 encoding columns too, so a wrong pitch or offset anywhere shows."""
    from e3dge_amd.local_query import Fuse_sft_MLP, local_features_from_maps
    DEV = "cuda:0"
    torch.manual_seed(5)
    B, H, S, C = 2, 8, 6, 256
    fuse = Fuse_sft_MLP(C + 1, C).to(DEV)
    for p_ in fuse.parameters():
        torch.nn.init.normal_(p_, std=0.05)
    maps = {'ref': torch.randn(B, C, 16, 16, device=DEV).requires_grad_(True), 'que': torch.randn(B, C, 24, 24, device=DEV).requires_grad_(True)}
    calib = torch.tensor([[[2.0, 0.0, 0.0, 0.0], [0.0, 2.0, 0.0, 0.0], [0.0, 0.0, 1.0, 2.0]]], device=DEV).repeat(B, 1, 1)
    batch = dict(feature_maps=maps, ref_calibs=calib, que_calibs=calib.clone(), points=(torch.rand(B, H, H, S, 3, device=DEV) - 0.5),
                 xyz=(torch.rand(B, 3, H, H, device=DEV) - 0.5), fuse_sft_block=fuse)
    gw = torch.randn(B, H, H, S, 301, device=DEV)

    def run():
        for t_ in list(maps.values()) + list(fuse.parameters()):
            t_.grad = None
        f, m = local_features_from_maps(batch)
        (f * gw).sum().backward()
        return f.detach().clone(), m.clone(), [maps['ref'].grad.clone(), maps['que'].grad.clone()] + [p_.grad.clone() for p_ in fuse.parameters()], f.grad_fn
    f1, m1, g1, fn1 = run()
    monkeypatch.setenv("E3DGE_LOCAL_FEATS_NODES", "composed")
    f0, m0, g0, fn0 = run()
    assert "FuseFn" in type(fn1).__name__ or "FuseFn" in str(fn1.next_functions), type(fn1).__name__
    assert torch.equal(f1, f0) and torch.equal(m1, m0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (a.shape, float((a - b).abs().max()), float(b.abs().max()))
