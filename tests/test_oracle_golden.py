"""CPU: the oracle restatement (oracle/*.py) against the golden vectors recorded from the REAL reference
(oracle/gen_golden.py).  This is what pins the oracle; it runs everywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest
import torch

from conftest import full_state_dict, load_golden
from oracle import camera_ref, decoder_ref, ops_ref, renderer_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn

T = torch.from_numpy
RENDER_KEYS = ['rays_d', 'dists', 'hit_prob', 'points', 'sdf', 'gen_thumb_imgs', 'features', 'mask', 'xyz', 'depth',
               'viewdirs']


@pytest.fixture(scope="module")
def sd():
    return full_state_dict()[1]


def _close(a, b, atol, rtol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) - rtol * np.abs(b)
    assert err.max() <= atol, f"max err {np.abs(a - b).max():.3e} > {atol:.1e}"


@pytest.mark.parametrize("name,sub", [("renderer_16x24", None), ("renderer_8x48", None), ("renderer_8x18", None),
                                      ("renderer_64x24", 8)])
def test_renderer_restatement_matches_reference(sd, name, sub):
    g = load_golden(name)
    B, res, S = int(g['batch']), int(g['res']), int(g['n_samples'])
    wr, _ = syn.synthetic_inputs(B, seed=int(g['styles_seed']))
    with torch.no_grad():
        out = renderer_ref.render(sd, T(g['poses']), T(g['focal']), T(g['near']), T(g['far']), wr, res=res, n_samples=S)
    for k in RENDER_KEYS:
        v = out[k]
        if sub and k in ('sdf', 'hit_prob', 'points', 'dists'):
            v = v[:, ::sub, ::sub]
        if sub and k == 'features':
            v = v[:, :, ::sub, ::sub]
        # same PyTorch build + same algorithm -> essentially bit-identical; 1e-6 absorbs thread-count effects
        _close(v.numpy(), g['ref_' + k], atol=2e-6, rtol=1e-6)
        assert abs(float(out[k].double().sum()) - float(g['sum_' + k])) <= 1e-4 * max(1.0, abs(float(g['sum_' + k])))


def test_float64_truth_is_reproducible(sd):
    g = load_golden("renderer_8x48")
    wr, _ = syn.synthetic_inputs(1, seed=1)
    with torch.no_grad():
        out = renderer_ref.render(sd, T(g['poses']), T(g['focal']), T(g['near']), T(g['far']), wr, res=8, n_samples=48,
                                  dtype=torch.float64)
    _close(out['features'].numpy(), g['f64_features'], atol=1e-6)
    _close(out['sdf'].numpy(), g['f64_sdf'], atol=1e-6)


def test_point_queries_and_film(sd):
    g = load_golden("points")
    wr, _ = syn.synthetic_inputs(2, seed=1)
    with torch.no_grad():
        raw0 = renderer_ref.query_points(sd, T(g['pts']), None, wr)
        raw1 = renderer_ref.query_points(sd, T(g['pts']), T(g['viewdirs']), wr)
        film = renderer_ref.film_params(sd, 'renderer.network.', wr)
    _close(raw0.numpy(), g['ref_raw_zero_view'], atol=2e-6)
    _close(raw1.numpy(), g['ref_raw_view'], atol=2e-6)
    _close(film.numpy(), g['ref_film'], atol=1e-5)


def test_texture_film_path(sd):
    g = load_golden("renderer_tex_8x24")
    wr, _ = syn.synthetic_inputs(1, seed=1)
    tex = syn.synthetic_tex_conditions(1, 8, 24, seed=int(g['tex_seed']))
    with torch.no_grad():
        out = renderer_ref.render(sd, T(g['poses']), T(g['focal']), T(g['near']), T(g['far']), wr, res=8, n_samples=24, tex=tex)
    _close(out['gen_thumb_imgs'].numpy(), g['ref_gen_thumb_imgs'], atol=2e-6)
    _close(out['features'].numpy(), g['ref_features'], atol=2e-6)
    _close(out['sdf'].numpy(), g['ref_sdf'], atol=2e-6)      # the tex FiLM must not touch the geometry head
    _close(out['hit_prob'].numpy(), g['ref_hit_prob'], atol=2e-6)


UPFIRDN_CASES = ['blur_up', 'upsample', 'downsample', 'blur_down', 'k3', 'crop', 'big']


@pytest.mark.parametrize("name", UPFIRDN_CASES)
def test_upfirdn2d_restatement(name):
    g = load_golden("upfirdn2d")
    up, down, p0, p1 = [int(v) for v in g[name + '_cfg']]
    x = T(g[name + '_x']).requires_grad_(True)
    y = ops_ref.upfirdn2d_ref_simple(x, T(g[name + '_k']), up, down, (p0, p1))
    _close(y.detach().numpy(), g[name + '_y'], atol=1e-6)
    gx, = torch.autograd.grad(y, x, T(g[name + '_gy']))
    _close(gx.numpy(), g[name + '_gx'], atol=1e-5)


def test_upfirdn2d_asymmetric():
    g = load_golden("upfirdn2d")
    ux, uy, dx, dy, px0, px1, py0, py1 = [int(v) for v in g['asym_cfg']]
    y = ops_ref.upfirdn2d_ref(T(g['asym_x']), T(g['asym_k']), (ux, uy), (dx, dy), (px0, px1, py0, py1))
    _close(y.numpy(), g['asym_y'], atol=1e-6)


@pytest.mark.parametrize("name", ['conv', 'mapping', 'nobias', 'ragged'])
def test_fused_act_restatement(name):
    g = load_golden("fused_act")
    x = T(g[name + '_x']).requires_grad_(True)
    b = T(g[name + '_b']).requires_grad_(True) if (name + '_b') in g else None
    y = ops_ref.fused_leaky_relu_ref(x, b, 0.2, float(g[name + '_scale']))
    _close(y.detach().numpy(), g[name + '_y'], atol=1e-6)
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), T(g[name + '_gy']))
    _close(grads[0].numpy(), g[name + '_gx'], atol=1e-6)
    if b is not None:
        _close(grads[1].numpy(), g[name + '_gb'], atol=1e-4)
    # the raw op's act/grad table reproduces forward and backward of the same function
    fwd = ops_ref.fused_bias_act_ref(x.detach(), None if b is None else b.detach(), None, 3, 0, 0.2, float(g[name + '_scale']))
    _close(fwd.numpy(), g[name + '_y'], atol=1e-6)
    bwd = ops_ref.fused_bias_act_ref(T(g[name + '_gy']), None, fwd, 3, 1, 0.2, float(g[name + '_scale']))
    _close(bwd.numpy(), g[name + '_gx'], atol=1e-6)


def test_decoder_and_mappings(sd):
    g = load_golden("decoder_256")
    _, wd = syn.synthetic_inputs(1, seed=1)
    feats = T((0.5 * np.random.RandomState(int(g['feats_seed'])).standard_normal((1, 256, 64, 64))).astype(np.float32))
    with torch.no_grad():
        img = decoder_ref.decoder_forward(sd, feats, wd[:, :6])
        w = decoder_ref.renderer_mapping(sd, T(g['z']))
        wdec = decoder_ref.decoder_mapping(sd, w)
    _close(img.numpy(), g['ref_img'], atol=2e-5)
    _close(w.numpy(), g['ref_w'], atol=1e-5)
    _close(wdec.numpy(), g['ref_wdec'], atol=1e-4 * float(np.abs(g['ref_wdec']).max()))


def test_camera_restatement():
    g = load_golden("camera")
    poses, focal, near, far = camera_ref.camera_from_locations(64, T(g['locations']))
    _close(poses.numpy(), g['ref_poses'], atol=1e-6)
    _close(focal.numpy(), g['ref_focal'], atol=1e-4)
    _close(near.numpy(), g['ref_near'], atol=1e-7)
    _close(far.numpy(), g['ref_far'], atol=1e-7)
    poses_t, focal_t, _, _ = camera_ref.camera_from_locations(128, T(g['traj_locations']))
    _close(poses_t.numpy(), g['ref_traj_poses'], atol=1e-6)
    _close(focal_t.numpy(), g['ref_traj_focal'], atol=2e-4)


def test_training_direction_restatement_matches_reference_gradients():
    """The restatement's autograd (eikonal terms with create_graph, loss.backward to the styles) against the vectors
    recorded from the reference's own VolumeFeatureRenderer.forward + backward (oracle/gen_golden_grads.py)."""
    from oracle.training_ref import restated, stage1_loss
    g = load_golden("grads_8x18")
    sd = full_state_dict(res=int(g['res']), n_samples=int(g['n_samples']))[1]
    wr, _ = syn.synthetic_inputs(1, seed=int(g['styles_seed']))
    T = torch.from_numpy
    s = wr.clone().requires_grad_(True)
    o = restated(sd, T(g['poses']), T(g['focal']), T(g['near']), T(g['far']), s, T(g['uniform_pts']), T(g['surface_pts']),
                 int(g['res']), int(g['n_samples']), torch.float32)
    loss = stage1_loss(o, T(g['normals_gt']), T(g['g_feat']))
    loss.backward()
    assert np.abs(o['eikonal_term'].detach().numpy() - g['ref_eikonal_term']).max() <= 1e-6
    assert np.abs(o['xyz_rec_eikonal_term'].detach().numpy() - g['ref_xyz_rec_eikonal_term']).max() <= 1e-6
    assert np.abs(o['uniform_pts_rec'].detach().numpy() - g['ref_uniform_pts_rec']).max() <= 1e-7
    assert abs(float(loss) - float(g['ref_loss'])) <= 1e-6 * abs(float(g['ref_loss']))
    rel = np.abs(s.grad.numpy() - g['ref_dstyles']).max() / np.abs(g['ref_dstyles']).max()
    assert rel <= 1e-6, rel


def test_c5_restatement_matches_reference_gradients():
    """SURVEY.md 8d's C5 loss with the surface normal at the integrated point kept in the graph (:921-930) and a loss on
    hit_prob: the restatement's autograd against the reference's recorded step (tests/golden/grads_c5_8x18.npz)."""
    from oracle.training_ref import c5_loss, restated_c5
    g = load_golden("grads_c5_8x18")
    sd = full_state_dict(res=int(g['res']), n_samples=int(g['n_samples']))[1]
    wr, _ = syn.synthetic_inputs(1, seed=int(g['styles_seed']))
    T = torch.from_numpy
    for key, extra in (('', None), ('_hit', T(g['g_hit']))):
        s = wr.clone().requires_grad_(True)
        o = restated_c5(sd, T(g['poses']), T(g['focal']), T(g['near']), T(g['far']), s, int(g['res']), int(g['n_samples']), torch.float32)
        loss = c5_loss(o) if extra is None else c5_loss(o) + (o['hit_prob'] * extra).mean()
        loss.backward()
        assert abs(float(loss) - float(g['ref_loss' + key])) <= 1e-6 * abs(float(g['ref_loss' + key]))
        rel = np.abs(s.grad.numpy() - g['ref_dstyles' + key]).max() / np.abs(g['ref_dstyles' + key]).max()
        assert rel <= 5e-6, rel
    assert np.abs(o['surface_eikonal_term'].detach().numpy() - g['ref_surface_eikonal_term']).max() <= 1e-6
    # the path through the surface point is a large part of that term's gradient: detaching xyz changes it by ~50 %
    d = np.abs(g['f64_dstyles_surf_only'] - g['f64_dstyles_surf_only_detached_xyz']).max() / np.abs(g['f64_dstyles_surf_only']).max()
    assert d > 0.1


def test_texhead_restatement_matches_reference_class():
    """oracle/renderer_ref.tex_modulations against the vectors recorded from the reference's ResnetBlockFC
    (oracle/gen_golden_texhead.py)."""
    g = load_golden("texhead_301")
    prefix = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
    cin = int(g['cin'])
    sd = {prefix + k: syn.synthetic_tensor(prefix + k, shp) for k, shp in
          (('fc_0.weight', (cin, cin)), ('fc_0.bias', (cin,)), ('fc_1.weight', (512, cin)), ('fc_1.bias', (512,)),
           ('shortcut.weight', (512, cin)))}
    shp = tuple(int(v) for v in g['feats_shape'])
    feats = syn.synthetic_local_feats(shp[0], shp[1], shp[3], cin=cin, seed=int(g['feats_seed'])).reshape(shp)
    with torch.no_grad():
        a, b = renderer_ref.tex_modulations(sd, prefix, feats)
    assert np.abs(a.numpy() - g['ref_alpha']).max() == 0 and np.abs(b.numpy() - g['ref_beta']).max() == 0


@pytest.mark.parametrize("variant", ["wide", "s2", "x4", "x32"])
def test_stress_fixtures_are_reproduced_by_the_oracle(variant):
    """tests/golden/stress_*.npz (trained-like magnitudes, recorded from the reference by oracle/gen_golden_stress.py): the
    restatement reproduces the reference bit for bit there too, renderer and decoder."""
    g = load_golden(f"stress_{variant}")
    sd = syn.stress_state_dict(full_state_dict(size=256, cm=1, res=16)[1], variant)
    wr, wd = syn.stress_inputs(variant, 1, seed=int(g['styles_seed']))
    c = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        o = renderer_ref.render(sd, c('poses'), c('focal'), c('near'), c('far'), wr, res=16, n_samples=24)
        t = renderer_ref.render(sd, c('poses'), c('focal'), c('near'), c('far'), wr, res=16, n_samples=24, dtype=torch.float64)
        for k in ('sdf', 'gen_thumb_imgs', 'depth', 'hit_prob', 'xyz'):
            assert float((o[k] - c('ref_' + k)).abs().max()) == 0.0, k
        assert float((o['features'][:, ::4] - c('ref_features')).abs().max()) == 0.0
        if variant in ("wide", "x32"):
            img = decoder_ref.decoder_forward(sd, c('feats'), wd)
            assert float((img[:, :, ::2, ::2] - c('ref_img_sub2')).abs().max()) == 0.0
