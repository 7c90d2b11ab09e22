"""GPU: the local-feature -> texture-FiLM head (e3dge_tex_modulations_fwd) against the vectors recorded
from the reference's ResnetBlockFC and against the oracle on other sizes; then through the renderer's second pass.

Stated fp32 tolerance: outputs are O(10); the reference's own fp32 result is 9e-6 (abs) from the float64 evaluation on
the fixture.  Bound: |hip - reference| <= 2e-5 * max(1, max|out|/10) (measured 4-9e-6) and |hip - f64| <= 3x the fp32 oracle's own
distance (+1e-5)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import full_state_dict, load_golden, maxerr, record
from oracle import renderer_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.volume_renderer import ResnetBlockFC, VolumeFeatureRenderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PREFIX = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'


def make_head(cin):
    h = ResnetBlockFC(cin, 512)
    sd = {k: syn.synthetic_tensor(PREFIX + k, v.shape) for k, v in h.state_dict().items()}
    h.load_state_dict(sd)
    return h.to(DEV), {PREFIX + k: v for k, v in sd.items()}


def test_texhead_against_reference_golden():
    g = load_golden("texhead_301")
    cin = int(g['cin'])
    h, _ = make_head(cin)
    shp = tuple(int(v) for v in g['feats_shape'])
    feats = syn.synthetic_local_feats(shp[0], shp[1], shp[3], cin=cin, seed=int(g['feats_seed']), device=DEV).reshape(shp)
    with torch.no_grad():
        a, b = h.tex_modulations(feats)
        both = h(feats)
    assert tuple(a.shape) == shp[:-1] + (256,) and tuple(both.shape) == shp[:-1] + (512,)
    e = dict(alpha_vs_ref=maxerr(a, g['ref_alpha']), beta_vs_ref=maxerr(b, g['ref_beta']),
             alpha_vs_f64=maxerr(a, g['f64_alpha']), beta_vs_f64=maxerr(b, g['f64_beta']),
             ref_vs_f64=float(max(np.abs(g['ref_alpha'] - g['f64_alpha']).max(), np.abs(g['ref_beta'] - g['f64_beta']).max())))
    record("texhead_golden_301", **e)
    assert max(e["alpha_vs_ref"], e["beta_vs_ref"]) <= 2e-5, e
    assert max(e['alpha_vs_f64'], e['beta_vs_f64']) <= 3 * e['ref_vs_f64'] + 1e-5, e
    assert torch.equal(both[..., :256], a) and torch.equal(both[..., 256:], b)


@pytest.mark.parametrize("cin", [64, 256, 301, 320])
@pytest.mark.parametrize("n", [1, 127, 129, 5000])
def test_texhead_sizes_against_oracle(cin, n):
    h, sd = make_head(cin)
    rs = np.random.RandomState(cin + n)
    feats = torch.from_numpy((rs.standard_normal((n, cin)) * (0.1 + 2 * rs.uniform(size=(1, cin)))).astype(np.float32)).to(DEV)
    with torch.no_grad():
        a, b = h.tex_modulations(feats)
        ra, rb = renderer_ref.tex_modulations(sd, PREFIX, feats.cpu())
        ta, tb = renderer_ref.tex_modulations(sd, PREFIX, feats.cpu(), dtype=torch.float64)
    scale = max(1.0, float(ta.abs().max()) / 10)
    e = max(maxerr(a, ra), maxerr(b, rb))
    e64, o64 = max(maxerr(a, ta), maxerr(b, tb)), max(maxerr(ra, ta), maxerr(rb, tb))
    record(f"texhead_cin{cin}_n{n}", hip_vs_oracle=e, hip_vs_f64=e64, oracle_vs_f64=o64, out_max=float(ta.abs().max()))
    assert e <= 2e-5 * scale and e64 <= 3 * o64 + 1e-5 * scale, (e, e64, o64)


@pytest.mark.parametrize("cin", [1, 2, 3, 5, 7, 62, 299, 301, 302, 303, 317])
@pytest.mark.parametrize("n", [1, 2, 33])
def test_texhead_ragged_row_ends(cin, n):
    """Rows are read in 16-byte pieces at 4-byte alignment; a piece that sticks out of its row reads into the next row and
    is zeroed, and the pieces at the very end of the tensor are clamped and shifted back.  The tensor sits at an odd offset
    inside a NaN-filled buffer: any element taken from outside its row (or outside the tensor) poisons the output."""
    h, sd = make_head(cin)
    rs = np.random.RandomState(1000 * cin + n)
    x = (rs.standard_normal((n, cin)) * (0.1 + 2 * rs.uniform(size=(1, cin)))).astype(np.float32)
    buf = torch.full((3 + n * cin + 64,), float('nan'), device=DEV)
    buf[3:3 + n * cin] = torch.from_numpy(x).reshape(-1).to(DEV)
    feats = buf[3:3 + n * cin].view(n, cin)
    assert feats.is_contiguous() and feats.data_ptr() % 16 != 0
    with torch.no_grad():
        a, b = h.tex_modulations(feats)
        ra, rb = renderer_ref.tex_modulations(sd, PREFIX, torch.from_numpy(x))
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    scale = max(1.0, float(ra.abs().max()) / 10)
    e = max(maxerr(a, ra), maxerr(b, rb))
    record(f"texhead_ragged_cin{cin}_n{n}", hip_vs_oracle=e)
    assert e <= 2e-5 * scale, e


@pytest.mark.parametrize("mag", [1e-6, 1e+4])
def test_texhead_input_magnitude(mag):
    """Per-point block scaling of the operands: the relative error does not depend on the input's magnitude."""
    h, sd = make_head(301)
    rs = np.random.RandomState(3)
    feats = torch.from_numpy((mag * rs.standard_normal((700, 301))).astype(np.float32)).to(DEV)
    with torch.no_grad():
        a, b = h.tex_modulations(feats)
        ta, tb = renderer_ref.tex_modulations(sd, PREFIX, feats.cpu(), dtype=torch.float64)
    ref = torch.cat([ta, tb], -1)
    # the bias terms are O(0.05): compare relative to the output's own scale
    rel = float((torch.cat([a, b], -1).double().cpu() - ref).abs().max() / ref.abs().max())
    record(f"texhead_magnitude_{mag:g}", rel_err=rel)
    assert rel <= 5e-6, rel
    with torch.no_grad():
        assert torch.equal(*[h.tex_modulations(feats)[0] for _ in range(2)])
        empty = h.tex_modulations(torch.empty(0, 301, device=DEV))
    assert empty[0].shape == (0, 256)


def test_texhead_backward_against_f64_autograd():
    """Stage-2 training differentiates the texture head (e3dge_full_runner.py:185-317): gradients w.r.t. the local features
    and the five parameters against float64 autograd of the oracle.  Tolerance: 2e-5 of each gradient's maximum."""
    h, sd = make_head(301)
    rs = np.random.RandomState(5)
    feats = torch.from_numpy(rs.standard_normal((515, 301)).astype(np.float32)).to(DEV).requires_grad_(True)
    ga = torch.from_numpy(rs.standard_normal((515, 256)).astype(np.float32)).to(DEV)
    gb = torch.from_numpy(rs.standard_normal((515, 256)).astype(np.float32)).to(DEV)
    a, b = h.tex_modulations(feats)
    ((a * ga).sum() + (b * gb).sum()).backward()
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    f64 = feats.detach().cpu().double().requires_grad_(True)
    ta, tb = renderer_ref.tex_modulations(sd64, PREFIX, f64, dtype=torch.float64)
    ((ta * ga.cpu().double()).sum() + (tb * gb.cpu().double()).sum()).backward()
    errs = {'feats': float((feats.grad.cpu().double() - f64.grad).abs().max() / f64.grad.abs().max())}
    for name, p_ in h.named_parameters():
        t = sd64[PREFIX + name].grad
        errs[name] = float((p_.grad.cpu().double() - t).abs().max() / t.abs().max())
    record("texhead_backward", **errs)
    assert max(errs.values()) <= 2e-5, errs


def _head_grads(h, feats, ga, gb, params=False):
    f = feats.detach().clone().requires_grad_(True)
    for p_ in h.parameters():
        p_.requires_grad_(params)
        p_.grad = None
    a, b = h.tex_modulations(f)
    ((a * ga).sum() + (b * gb).sum()).backward()
    return f.grad, {n: p_.grad for n, p_ in h.named_parameters()} if params else {}


@pytest.mark.parametrize("cin,n", [(301, 1), (301, 127), (301, 129), (301, 1000), (320, 257), (64, 300), (5, 3), (3, 1), (1, 70)])
def test_texhead_data_gradient_kernel_sizes(cin, n, monkeypatch):
    """e3dge_tex_modulations_bwd (round 5) on ragged sizes: row ends that are not 16-byte aligned, fewer rows than a tile, the tensor's last rows
    (clamped reads), cin below one quad.  Against float64 autograd of the oracle (2e-5 of the gradient's maximum + the relu-branch allowance
    below) and against round 4's library chain on the same inputs."""
    h, sd = make_head(cin)
    rs = np.random.RandomState(cin + n)
    feats = torch.from_numpy(rs.standard_normal((n, cin)).astype(np.float32)).to(DEV)
    ga = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    gb = torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    g_hip, _ = _head_grads(h, feats, ga, gb)
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "library")
    g_lib, _ = _head_grads(h, feats, ga, gb)
    monkeypatch.delenv("E3DGE_TEXHEAD_BWD")
    sd64 = {k: v.double() for k, v in sd.items()}
    f64 = feats.cpu().double().requires_grad_(True)
    ta, tb = renderer_ref.tex_modulations(sd64, PREFIX, f64, dtype=torch.float64)
    ((ta * ga.cpu().double()).sum() + (tb * gb.cpu().double()).sum()).backward()
    scale = float(f64.grad.abs().max())
    e_hip = float((g_hip.cpu().double() - f64.grad).abs().max()) / scale
    e_lib = float((g_lib.cpu().double() - f64.grad).abs().max()) / scale
    record("texhead_bwd_sizes", cin=cin, n=n, hip_vs_f64=e_hip, library_vs_f64=e_lib)
    assert torch.isfinite(g_hip).all() and g_hip.shape == (n, cin)
    # a hidden unit within rounding of 0 may take the other relu branch than float64 does (the library chain has the same freedom): allow what
    # the fp32 library path shows on the same inputs
    assert e_hip <= max(2e-5, 2 * e_lib), (e_hip, e_lib)


@pytest.mark.parametrize("mag_x,mag_g", [(1e-3, 1e-6), (1.0, 1e4), (50.0, 1e-2), (1e3, 1.0)])
def test_texhead_data_gradient_magnitudes(mag_x, mag_g):
    """Block scaling of the three operands (x, d out, d net) over ten orders of magnitude, rows of very different size in one tile."""
    h, sd = make_head(301)
    rs = np.random.RandomState(17)
    n = 300
    row = torch.from_numpy(np.exp(rs.uniform(-6, 6, (n, 1))).astype(np.float32)).to(DEV)
    feats = mag_x * torch.from_numpy(rs.standard_normal((n, 301)).astype(np.float32)).to(DEV)
    ga = mag_g * row * torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    gb = mag_g * row * torch.from_numpy(rs.standard_normal((n, 256)).astype(np.float32)).to(DEV)
    g_hip, _ = _head_grads(h, feats, ga, gb)
    sd64 = {k: v.double() for k, v in sd.items()}
    f64 = feats.cpu().double().requires_grad_(True)
    ta, tb = renderer_ref.tex_modulations(sd64, PREFIX, f64, dtype=torch.float64)
    ((ta * ga.cpu().double()).sum() + (tb * gb.cpu().double()).sum()).backward()
    rel = ((g_hip.cpu().double() - f64.grad).abs().amax(1) / f64.grad.abs().amax(1).clamp_min(1e-300))       # per point: every row has its own scale
    record("texhead_bwd_magnitudes", mag_x=mag_x, mag_g=mag_g, worst_row=float(rel.max()), median_row=float(rel.median()))
    assert float(rel.median()) <= 2e-6 and float((rel > 2e-5).double().mean()) <= 0.02, (float(rel.max()), float(rel.median()))


def test_texhead_backward_zero_initialised_head_and_parameter_gradients(monkeypatch):
    """The reference initialises the head to zero (HGPIFuGANNetResidualInputResnetFC.py:88-93): d feats = 0 exactly, and the parameter gradients
    (library GEMMs on the kernel's d net) equal the library chain's."""
    h = ResnetBlockFC(301, 512).to(DEV)
    rs = np.random.RandomState(3)
    feats = torch.from_numpy(rs.standard_normal((200, 301)).astype(np.float32)).to(DEV)
    ga = torch.from_numpy(rs.standard_normal((200, 256)).astype(np.float32)).to(DEV)
    gb = torch.from_numpy(rs.standard_normal((200, 256)).astype(np.float32)).to(DEV)
    g0, p0 = _head_grads(h, feats, ga, gb, params=True)
    assert float(g0.abs().max()) == 0.0
    h2, _ = make_head(301)
    g_hip, p_hip = _head_grads(h2, feats, ga, gb, params=True)
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "library")
    g_lib, p_lib = _head_grads(h2, feats, ga, gb, params=True)
    errs = {"feats": float((g_hip - g_lib).abs().max() / g_lib.abs().max())}
    for k in p_lib:
        errs[k] = float((p_hip[k] - p_lib[k]).abs().max() / p_lib[k].abs().max().clamp_min(1e-30))
    record("texhead_bwd_hip_vs_library", **errs)
    assert max(errs.values()) <= 2e-5, errs
    for p_ in h2.parameters():
        p_.requires_grad_(True)


@pytest.mark.parametrize("P,m,n,relu", [(1, 5, 3, False), (31, 128, 128, True), (33, 129, 127, False), (1000, 301, 301, True),
                                          (4097, 256, 301, False), (20000, 512, 320, True), (300, 1, 600, False)])
def test_wgrad_against_float64(P, m, n, relu, monkeypatch):
    """e3dge_wgrad: a^T f(b) over ragged sizes (fewer rows than a step, block edges, rows that are not 16-byte aligned, more than 512 columns),
    with magnitudes ten decades apart between the operands; against float64 and against the library matmul (fp32).  Tolerance: 2e-6 of the
    result's maximum + what fp32 summation of P terms costs the library path on the same data; bit-identical on repetition."""
    from e3dge_amd.wgrad import wgrad
    rs = np.random.RandomState(P + m)
    a = torch.from_numpy((1e-5 * rs.standard_normal((P, m + 3))).astype(np.float32)).to(DEV)[:, 1:m + 1]          # a view: pitch m + 3, offset 1
    b = torch.from_numpy((3e4 * rs.standard_normal((P, n))).astype(np.float32)).to(DEV)
    c = wgrad(a, b, relu_b=relu)
    c2 = wgrad(a, b, relu_b=relu)
    ref = a.double().t() @ (torch.relu(b.double()) if relu else b.double())
    lib = a.t() @ (torch.relu(b) if relu else b)
    scale = float(ref.abs().max())
    e_hip, e_lib = float((c.double() - ref).abs().max()) / scale, float((lib.double() - ref).abs().max()) / scale
    record("wgrad_vs_f64", P=P, m=m, n=n, relu=int(relu), hip=e_hip, library=e_lib)
    assert c.shape == (m, n) and torch.equal(c, c2)
    assert e_hip <= 2e-6 + 2 * e_lib, (e_hip, e_lib)
    out = torch.full((m, n + 5), 7.0, device=DEV)
    wgrad(a, b, relu_b=relu, out=out[:, 2:n + 2])
    assert torch.equal(out[:, 2:n + 2], c) and float(out[:, :2].min()) == 7.0 and float(out[:, n + 2:].min()) == 7.0
    # round 6: the bias gradient (column sums of a) from the same launch; the matrix itself must not change by a bit
    c3, cs = wgrad(a, b, relu_b=relu, colsum=True)
    e_cs = float((cs.double() - a.double().sum(0)).abs().max() / a.double().sum(0).abs().max().clamp_min(1e-300))
    e_cs_lib = float((a.sum(0).double() - a.double().sum(0)).abs().max() / a.double().sum(0).abs().max().clamp_min(1e-300))
    record("wgrad_colsum_vs_f64", P=P, m=m, hip=e_cs, library=e_cs_lib)
    assert torch.equal(c3, c) and cs.shape == (m,) and e_cs <= 2e-6 + 2 * e_cs_lib, (e_cs, e_cs_lib)


@pytest.mark.parametrize("P,m,relu", [(700, 256, True), (5000, 256, False), (98304, 256, True), (333, 40, False)])
def test_wgrad_with_the_mask_column_left_out_of_the_block_grid(P, m, relu):
    """Round 6: Fuse_sft_MLP's 513-wide input (256 features | visibility mask | 256 features): `gap_col=256` contracts 512 columns on the
    matrix pipe and the mask column as a weighted column sum in the same launch.  Against float64, and the 512 regular columns against
    two plain calls on the two halves (other slab counts: equal to rounding)."""
    from e3dge_amd.wgrad import wgrad
    rs = np.random.RandomState(P)
    a = torch.from_numpy(rs.standard_normal((P, m)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.standard_normal((P, 513)).astype(np.float32)).to(DEV)
    b[:, 256] = (b[:, 256] > 0).float()
    c, cs = wgrad(a, b, relu_b=relu, colsum=True, gap_col=256)
    ref = a.double().t() @ (torch.relu(b.double()) if relu else b.double())
    lib = a.t() @ (torch.relu(b) if relu else b)
    scale = float(ref.abs().max())
    e_hip, e_lib = float((c.double() - ref).abs().max()) / scale, float((lib.double() - ref).abs().max()) / scale
    e_col = float((c[:, 256].double() - ref[:, 256]).abs().max()) / float(ref[:, 256].abs().max())
    record("wgrad_gap_col_vs_f64", P=P, m=m, hip=e_hip, library=e_lib, mask_column=e_col)
    assert e_hip <= 2e-6 + 2 * e_lib and e_col <= 2e-6 + 2 * e_lib, (e_hip, e_col, e_lib)
    halves = torch.cat([wgrad(a, b[:, :256], relu_b=relu), wgrad(a, b[:, 257:], relu_b=relu)], 1)
    both = torch.cat([c[:, :256], c[:, 257:]], 1)
    assert float((both - halves).abs().max()) <= 2e-6 * scale
    assert float((cs.double() - a.double().sum(0)).abs().max()) <= 1e-5 * float(a.double().sum(0).abs().max())


def test_texhead_parameter_gradients_native_vs_library(monkeypatch):
    """The head trainable: the five parameter gradients from e3dge_wgrad on the backward kernel's d net / net against the library chain."""
    h, _ = make_head(301)
    rs = np.random.RandomState(9)
    feats = torch.from_numpy(rs.standard_normal((3000, 301)).astype(np.float32)).to(DEV)
    ga = torch.from_numpy(rs.standard_normal((3000, 256)).astype(np.float32)).to(DEV)
    gb = torch.from_numpy(rs.standard_normal((3000, 256)).astype(np.float32)).to(DEV)
    g_hip, p_hip = _head_grads(h, feats, ga, gb, params=True)
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "library")
    g_lib, p_lib = _head_grads(h, feats, ga, gb, params=True)
    errs = {"feats": float((g_hip - g_lib).abs().max() / g_lib.abs().max())}
    for k in p_lib:
        errs[k] = float((p_hip[k] - p_lib[k]).abs().max() / p_lib[k].abs().max())
    record("texhead_param_grads_hip_vs_library", **errs)
    assert max(errs.values()) <= 2e-5, errs


def test_second_pass_from_local_feats():
    """VolumeFeatureRenderer.forward with local_data_batch={'feats': ...} (the reference's second pass with already
    queried local features, :434-437 + :327-336 + :217-220) == the oracle's render with the oracle's (alpha, beta)."""
    res, S = 8, 24
    g, sd = full_state_dict(res=res, n_samples=S)
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True),
                              out_im_res=res, mode='test')
    own = {}
    for k in r.state_dict():
        if 'netLocal' in k:
            own[k] = syn.synthetic_tensor('renderer.' + k, r.state_dict()[k].shape) * 0.05     # small FiLM perturbation
        else:
            own[k] = sd['renderer.' + k.replace('network.netGlobal.', 'network.')]
    r.load_state_dict(own)
    r = r.to(DEV)
    sd_all = dict(sd)
    sd_all.update({'renderer.' + k: v for k, v in own.items() if 'netLocal' in k})
    wr, _ = syn.synthetic_inputs(1, seed=2, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=torch.tensor([[0.2, 0.0]], device=DEV))
    feats = syn.synthetic_local_feats(1, res, S, device=DEV)
    with torch.no_grad():
        out = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        ta, tb = renderer_ref.tex_modulations(sd_all, PREFIX, feats.cpu())
        c = lambda t: t.detach().cpu()
        ref = renderer_ref.render(sd, c(poses), c(focal), c(near), c(far), c(wr), res=res, n_samples=S, tex=(ta, tb))
        plain = renderer_ref.render(sd, c(poses), c(focal), c(near), c(far), c(wr), res=res, n_samples=S)
    e = dict(features=maxerr(out['features'], ref['features']), rgb=maxerr(out['gen_thumb_imgs'], ref['gen_thumb_imgs']),
             sdf=maxerr(out['sdf'], ref['sdf']), tex_effect=maxerr(ref['features'], plain['features']))
    record("second_pass_from_local_feats", **e)
    assert e['tex_effect'] > 1e-2                        # the modulation does something
    assert e['features'] <= 1e-4 and e['rgb'] <= 5e-6 and e['sdf'] <= 1e-5, e


@pytest.mark.parametrize("res,S,B", [(8, 24, 1), (16, 18, 2), (16, 48, 1), (64, 24, 1)])
def test_second_pass_reads_the_first_passs_backbone(res, S, B, monkeypatch):
    """The second pass of an evaluated image (same styles / poses, texture FiLM behind the sdf head) reads the layer-7 record
    and the composite weights the first pass left behind (e3dge_siren_render_fwd backbone_out / backbone_in) instead of
    recomputing layers 0..7 + sdf head + transmittance scan: every output must be BIT-identical to a full second pass; a
    changed latent / pose / weight, grad mode, or E3DGE_REUSE_BACKBONE=0 must fall back to the full launch."""
    from e3dge_amd import volume_renderer as vr
    g, sd = full_state_dict(res=res, n_samples=S)
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True),
                              out_im_res=res, mode='test')
    own = {k: (syn.synthetic_tensor('renderer.' + k, v.shape) * 0.05 if 'netLocal' in k else
               sd['renderer.' + k.replace('network.netGlobal.', 'network.')]) for k, v in r.state_dict().items()}
    r.load_state_dict(own)
    r = r.to(DEV).eval()
    wr, _ = syn.synthetic_inputs(B, seed=4, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=0.2 * torch.randn(B, 2, device=DEV))
    feats = syn.synthetic_local_feats(B, res, S, device=DEV)
    launches = []
    orig = r.render_with_film

    def spy(*a, **k):
        out = orig(*a, **k)
        rec = vr._BACKBONE.get(r)                             # a launch that started from the record returns the first pass's tensors
        tex = (a[5] if len(a) > 5 else k.get('tex_conditions')) is not None
        launches.append((tex, tex and rec is not None and out['sdf'] is rec['out']['sdf']))
        return out
    monkeypatch.setattr(r, "render_with_film", spy)
    keys = ('features', 'gen_thumb_imgs', 'sdf', 'hit_prob', 'xyz', 'depth', 'mask', 'points', 'dists', 'rays_d', 'viewdirs')
    with torch.no_grad():
        monkeypatch.setenv("E3DGE_REUSE_BACKBONE", "0")
        r(poses, focal, near, far, styles=wr)
        full = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert vr._BACKBONE.get(r) is None
        monkeypatch.setenv("E3DGE_REUSE_BACKBONE", "1")
        launches.clear()
        p1 = r(poses, focal, near, far, styles=wr)
        fast = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        again = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': 0.5 * feats})     # the record serves any number of second passes
        assert launches == [(False, False), (True, True), (True, True)], launches
        for k in keys:
            assert torch.equal(fast[k], full[k]), k
        assert not torch.equal(again['features'], fast['features']) and torch.equal(again['sdf'], full['sdf'])
        assert torch.equal(p1['sdf'], full['sdf'])
        # misses: another latent (new tensor), the same tensor edited in place, another pose
        launches.clear()
        wr2 = wr.clone()
        miss1 = r(poses, focal, near, far, styles=wr2, local_data_batch={'feats': feats})
        wr.mul_(1.0)
        miss2 = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert [l[1] for l in launches] == [False, False], launches
        for k in keys:
            assert torch.equal(miss1[k], full[k]) and torch.equal(miss2[k], full[k]), k
        r(poses, focal, near, far, styles=wr)
        r.invalidate()                                     # (what a caller does after editing weights / inputs through .data)
        launches.clear()
        miss3 = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert launches == [(True, False)] and torch.equal(miss3['features'], full['features'])
        r(poses, focal, near, far, styles=wr)
        poses2 = poses.clone()
        poses2[:, 0, 3] += 0.05
        launches.clear()
        other = r(poses2, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert launches == [(True, False)] and not torch.equal(other['xyz'], full['xyz'])
    # the C entry refuses inconsistent requests
    from e3dge_amd import _lib
    lib = _lib.load()
    one = 16
    bad = _lib.RenderArgs(packed=one, film=one, c2w=one, focal=one, near=one, far=one, t_vals=one, sigmoid_beta=1.0, batch=1, height=8, width=8,
                          n_samples=24, res=8, precision=_lib.PREC_F32, backbone_out=one)
    assert lib.e3dge_siren_render_fwd(ctypes.byref(bad), None) == -1                      # only the f16x3 kernel has the hand-over
    bad.precision, bad.backbone_out, bad.backbone_in = _lib.PREC_F16X3, None, one
    assert lib.e3dge_siren_render_fwd(ctypes.byref(bad), None) == -1                      # backbone_in without weights_in
    assert lib.e3dge_siren_backbone_bytes(1, 64, 64, 24) == 256 * 3 * 8 * 16384 and lib.e3dge_siren_backbone_bytes(1, 8, 8, 4) == 0


def _local_renderer(res, S):
    g, sd = full_state_dict(res=res, n_samples=S)
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True),
                              out_im_res=res, mode='test')
    r.load_state_dict({k: (syn.synthetic_tensor('renderer.' + k, v.shape) * 0.05 if 'netLocal' in k else
                           sd['renderer.' + k.replace('network.netGlobal.', 'network.')]) for k, v in r.state_dict().items()})
    return r.to(DEV).eval()


def test_backbone_record_does_not_alias_a_recycled_latent(monkeypatch):
    """ABA guard (round-3 review): a latent that is freed and a NEW latent of the same size that the caching allocator
    places at the same address (both with version 0) must not be mistaken for each other.  The record holds the tensors
    it was keyed on, so their storage cannot be recycled while the record is alive; once the record is gone (a plain
    render that leaves none, or invalidate()) a second pass without its own first pass must be a full launch."""
    from e3dge_amd import volume_renderer as vr
    res, S = 16, 24
    r = _local_renderer(res, S)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=torch.zeros(1, 2, device=DEV))
    feats = syn.synthetic_local_feats(1, res, S, device=DEV)
    launches = []
    orig = r.render_with_film

    def spy(*a, **k):
        out = orig(*a, **k)
        rec = vr._BACKBONE.get(r)
        tex = (a[5] if len(a) > 5 else k.get('tex_conditions')) is not None
        launches.append(bool(tex and rec is not None and out['sdf'] is rec['out']['sdf']))
        return out
    monkeypatch.setattr(r, "render_with_film", spy)
    w_b_host = syn.synthetic_inputs(1, seed=12, device='cpu')[0]
    with torch.no_grad():
        w_a = syn.synthetic_inputs(1, seed=11, device=DEV)[0].clone()
        r(poses, focal, near, far, styles=w_a)
        ptr_a = w_a.data_ptr()
        del w_a                                                        # the record still holds it: the block is NOT returned
        w_b = torch.empty_like(w_b_host, device=DEV).copy_(w_b_host)
        assert w_b.data_ptr() != ptr_a, "the record must keep the latent's storage alive"
        got = r(poses, focal, near, far, styles=w_b, local_data_batch={'feats': feats})
        assert launches[-1] is False
        # the ABA shape proper: no record alive, latent freed, a new one lands on its address, tex pass without a pass #1
        r.invalidate()
        w_c = syn.synthetic_inputs(1, seed=11, device=DEV)[0].clone()
        r(poses, focal, near, far, styles=w_c)
        ptr_c = w_c.data_ptr()
        r(poses, focal, near, far, styles=w_c, return_eikonal=True)    # a plain render that cannot leave a record drops the old one
        assert vr._BACKBONE.get(r) is None
        del w_c
        w_d = torch.empty_like(w_b_host, device=DEV).copy_(w_b_host)
        recycled = w_d.data_ptr() == ptr_c
        launches.clear()
        got2 = r(poses, focal, near, far, styles=w_d, local_data_batch={'feats': feats})
        assert launches == [False]
        monkeypatch.setenv("E3DGE_REUSE_BACKBONE", "0")
        want = r(poses, focal, near, far, styles=w_d, local_data_batch={'feats': feats})
        for k in ('features', 'gen_thumb_imgs', 'sdf', 'hit_prob', 'xyz', 'depth'):
            assert torch.equal(got2[k], want[k]) and torch.equal(got[k], want[k]), k
    record("backbone_aba_guard", allocator_recycled_the_block=bool(recycled))


def test_backbone_record_notices_an_edited_first_pass_output():
    """The hit path hands the first pass's geometry tensors (and `weights_in`) to the second pass: an in-place edit of one
    of them in between must turn the hit into a full launch, not into a composite over edited weights.  The record does not
    keep the first pass's rgb / features alive."""
    from e3dge_amd import volume_renderer as vr
    res, S = 8, 24
    r = _local_renderer(res, S)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=torch.zeros(1, 2, device=DEV))
    feats = syn.synthetic_local_feats(1, res, S, device=DEV)
    wr = syn.synthetic_inputs(1, seed=3, device=DEV)[0]
    with torch.no_grad():
        p1 = r(poses, focal, near, far, styles=wr)
        assert set(vr._BACKBONE[r]['out']).isdisjoint({'rgb', 'features'})
        ok = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert ok['sdf'] is p1['sdf']                                  # hit: geometry tensors are the first pass's
        keep = ok['features'].clone()
        p1['hit_prob'].mul_(0.5)                                       # the caller scribbles over the weights
        again = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert again['sdf'] is not p1['sdf']                           # full launch
        assert torch.equal(again['features'], keep)


@pytest.mark.parametrize("res,S,B", [(8, 24, 1), (16, 18, 2), (16, 21, 1), (64, 24, 1)])
def test_head_and_film_in_one_launch_is_bit_identical(res, S, B, monkeypatch):
    """SURVEY 8 f1 as specified: on the record path the texture head applies (alpha + 1) h8 + beta itself and hands pass #2 the
    FiLM-ed layer-7 record (e3dge_tex_film_fwd) -- (alpha, beta) never reach HBM.  Every output must equal, bit for bit, the
    two-launch form (head -> (alpha, beta) -> FiLM inside the render kernel) and the full second pass; sizes include tiles
    whose point count is not a multiple of the 128-point sub-tile (S = 18, 21) and a batch of two."""
    r = _local_renderer(res, S)
    head = r.network.netLocal.local_feat_to_tex_modulations_linear
    wr, _ = syn.synthetic_inputs(B, seed=7, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=0.2 * torch.randn(B, 2, device=DEV))
    feats = syn.synthetic_local_feats(B, res, S, device=DEV)
    calls = []
    orig = head.tex_film
    monkeypatch.setattr(head, "tex_film", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    keys = ('features', 'gen_thumb_imgs', 'sdf', 'hit_prob', 'xyz', 'depth', 'mask')
    with torch.no_grad():
        monkeypatch.setenv("E3DGE_REUSE_BACKBONE", "0")
        full = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        monkeypatch.setenv("E3DGE_REUSE_BACKBONE", "1")
        monkeypatch.setenv("E3DGE_FUSE_TEXFILM", "0")
        r(poses, focal, near, far, styles=wr)
        two = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        assert not calls
        monkeypatch.setenv("E3DGE_FUSE_TEXFILM", "1")
        r(poses, focal, near, far, styles=wr)
        one = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats})
        again = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': feats.clone()})
        assert len(calls) == 2
        for k in keys:
            assert torch.equal(one[k], two[k]) and torch.equal(one[k], full[k]) and torch.equal(again[k], full[k]), k
        # a second pass without a record (another latent) falls back to (alpha, beta)
        calls.clear()
        wr2 = wr.clone()
        miss = r(poses, focal, near, far, styles=wr2, local_data_batch={'feats': feats})
        assert not calls and torch.equal(miss['features'], full['features'])
    # with a graph wanted, the head stays differentiable
    f2 = feats.clone().requires_grad_(True)
    for p_ in r.parameters():
        p_.requires_grad_(False)
    out = r(poses, focal, near, far, styles=wr, local_data_batch={'feats': f2})
    out['features'].square().mean().backward()
    assert f2.grad is not None and torch.isfinite(f2.grad).all() and float(f2.grad.abs().max()) > 0


def test_backward_launch_reports_the_operand_maxima():
    """Round 6: e3dge_tex_modulations_bwd writes the four amax buffers e3dge_wgrad needs (max |x|, |[d alpha | d beta]|, |d net|, |net|) from the
    values it holds anyway -- they must equal the maxima of the tensors it read / left behind, exactly (3,000 points: a ragged last sub-tile)."""
    h, _ = make_head(301)
    rs = np.random.RandomState(21)
    n = 3000
    x = torch.from_numpy((3.0 * rs.standard_normal((n, 301))).astype(np.float32)).to(DEV)
    ga = torch.from_numpy((1e-3 * rs.standard_normal((n, 256))).astype(np.float32)).to(DEV)
    gb = torch.from_numpy((2e-3 * rs.standard_normal((n, 256))).astype(np.float32)).to(DEV)
    dx, dnet, net, am = h._launch_bwd(x, ga, gb, want_net=True, want_amax=True)
    torch.cuda.synchronize()
    got = [float(am[k].max()) for k in range(4)]
    want = [float(x.abs().max()), float(torch.maximum(ga.abs().max(), gb.abs().max())), float(dnet.abs().max()), float(net.abs().max())]
    assert got == want, (got, want)
    assert all(v > 0 for v in want)
