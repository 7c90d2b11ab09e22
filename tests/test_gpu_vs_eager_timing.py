"""GPU: the HIP renderer against the oracle's restatement executed by eager PyTorch ON THE SAME MI355X (rocBLAS GEMMs +
elementwise kernels + autograd) -- i.e. what the reference's own code path costs on this hardware -- for the
inference render (C2) and for the training direction (forward + backward to the styles, the renderer part of C5).
Records both timings in gpurun_out/parity_report.jsonl; asserts that the results agree and warns if the fused path is not the
faster one.  The oracle is used as checker / comparison only."""
import time

import pytest
import torch

from conftest import full_state_dict, record
from oracle import renderer_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from test_gpu_renderer import make_renderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MARGIN = 1.5      # wall-clock comparisons on a possibly shared GPU: generous, the measured ratios are recorded; a miss WARNS
                  # (numbers in the parity report) instead of failing -- timing is bench.py's business, parity is this file's


def _slower(what, t):
    import warnings
    warnings.warn(f"{what}: fused path not faster than eager PyTorch on this box: {t}")


def timed(fn, n=10, warm=3, rounds=3):
    """Best of `rounds` averages over n calls (a single disturbed round must not decide an assertion)."""
    for _ in range(warm):
        fn()
    best = float("inf")
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


@pytest.mark.parametrize("batch", [1, 4])
def test_fused_vs_eager_same_gpu(batch):
    res, S = 64, 24
    sd = full_state_dict()[1]
    sd_dev = {k: v.to(DEV) for k, v in sd.items()}
    r = make_renderer(sd, res, S)
    wr, _ = syn.synthetic_inputs(batch, seed=7, device=DEV)
    loc = torch.linspace(-0.3, 0.3, batch, device=DEV).reshape(-1, 1).repeat(1, 2) * torch.tensor([1.0, 0.4], device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, batch=batch, locations=loc)
    G = torch.randn(batch, 256, res, res, device=DEV)
    G_rgb = torch.randn(batch, 3, res, res, device=DEV)

    def hip_fwd():
        with torch.no_grad():
            return r(poses, focal, near, far, styles=wr)

    def eager_fwd():
        with torch.no_grad():
            return renderer_ref.render(sd_dev, poses, focal, near, far, wr, res=res, n_samples=S)

    def train(render_fn):
        s = wr.clone().requires_grad_(True)
        out = render_fn(s)
        ((out['features'] * G).sum() + (out['gen_thumb_imgs'] * G_rgb).sum()).backward()
        return s.grad

    hip_train = lambda: train(lambda s: r(poses, focal, near, far, styles=s))
    eager_train = lambda: train(lambda s: renderer_ref.render(sd_dev, poses, focal, near, far, s, res=res, n_samples=S))

    a, b = hip_fwd(), eager_fwd()
    assert float((a['features'] - b['features']).abs().max()) <= 1e-4
    ga, gb = hip_train(), eager_train()
    rel = float((ga - gb).abs().max() / gb.abs().max())
    t = dict(hip_fwd_ms=timed(hip_fwd), eager_fwd_ms=timed(eager_fwd), hip_train_ms=timed(hip_train),
             eager_train_ms=timed(eager_train))
    rays = batch * res * res
    record(f"fused_vs_eager_same_gpu_b{batch}", grad_rel_diff=rel, **t,
           hip_fwd_rays_per_s=rays / t['hip_fwd_ms'] * 1e3, eager_fwd_rays_per_s=rays / t['eager_fwd_ms'] * 1e3,
           hip_train_rays_per_s=rays / t['hip_train_ms'] * 1e3, eager_train_rays_per_s=rays / t['eager_train_ms'] * 1e3)
    assert rel <= 2e-3      # two fp32 evaluations of an ill-conditioned sum; each is checked against float64 elsewhere
    # timing is recorded, not gated tightly: a shared GPU may disturb a round; the fused path is normally 2-7x faster
    if not (t['hip_fwd_ms'] < MARGIN * t['eager_fwd_ms'] and t['hip_train_ms'] < MARGIN * t['eager_train_ms']):
        _slower("renderer forward / training step", t)


def test_stage1_step_fused_vs_eager_same_gpu():
    """C5 (renderer part): 64x64x18, eikonal term over all ray samples + loss on it (double backward), eager = the
    restatement with autograd.grad(create_graph=True) exactly as the reference does it, on the same GPU."""
    res, S = 64, 18
    sd = full_state_dict(res=res, n_samples=S)[1]
    sd_dev = {k: v.to(DEV) for k, v in sd.items()}
    r = make_renderer(sd, res, S)
    wr, _ = syn.synthetic_inputs(1, seed=7, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=torch.zeros(1, 2, device=DEV))

    def loss_of(rgb, eik):
        return (rgb ** 2).mean() + ((eik.norm(dim=-1) - 1) ** 2).mean()

    def hip_step():
        s = wr.clone().requires_grad_(True)
        o = r(poses, focal, near, far, styles=s, return_eikonal=True)
        loss_of(o['gen_thumb_imgs'], o['eikonal_term']).backward()
        return s.grad

    def eager_step():
        s = wr.clone().requires_grad_(True)
        ro = renderer_ref.render(sd_dev, poses, focal, near, far, s, res=res, n_samples=S)
        x = ro['points'].detach().clone().requires_grad_(True)
        raw = renderer_ref.query_points(sd_dev, x, None, s)
        eik = torch.autograd.grad(raw[..., 3:4], x, torch.ones_like(raw[..., 3:4]), create_graph=True)[0]
        loss_of(ro['gen_thumb_imgs'], eik).backward()
        return s.grad

    ga, gb = hip_step(), eager_step()
    rel = float((ga - gb).abs().max() / gb.abs().max())
    t = dict(hip_ms=timed(hip_step, n=5, warm=2), eager_ms=timed(eager_step, n=5, warm=2))
    record("stage1_fused_vs_eager_same_gpu", grad_rel_diff=rel, **t, speedup=t['eager_ms'] / t['hip_ms'])
    assert rel <= 2e-3
    if not t['hip_ms'] < MARGIN * t['eager_ms']:
        _slower("stage-1 step", t)


def test_texhead_fused_vs_eager_same_gpu():
    """Second-pass texture head at C2 size (98,304 points x 301 features): one fused launch vs the restatement's three
    GEMMs + elementwise kernels in eager PyTorch on the same GPU."""
    from e3dge_amd.volume_renderer import ResnetBlockFC
    prefix = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
    h = ResnetBlockFC(301, 512)
    sd = {k: syn.synthetic_tensor(prefix + k, v.shape) for k, v in h.state_dict().items()}
    h.load_state_dict(sd)
    h = h.to(DEV)
    sd_dev = {prefix + k: v.to(DEV) for k, v in sd.items()}
    feats = syn.synthetic_local_feats(1, 64, 24, device=DEV)

    def hip():
        with torch.no_grad():
            return h.tex_modulations(feats)

    def eager():
        with torch.no_grad():
            return renderer_ref.tex_modulations(sd_dev, prefix, feats)
    a, b = hip(), eager()
    err = float(max((a[0] - b[0]).abs().max(), (a[1] - b[1]).abs().max()))
    t = dict(hip_ms=timed(hip), eager_ms=timed(eager))
    flops = 2 * (301 * 301 + 2 * 301 * 512) * feats.shape[1] * feats.shape[2] * feats.shape[3]
    record("texhead_fused_vs_eager_same_gpu", max_abs_diff=err, **t, speedup=t['eager_ms'] / t['hip_ms'],
           hip_algorithmic_tflops=flops / t['hip_ms'] / 1e9)
    assert err <= 5e-5, (err, t)
    if not t['hip_ms'] < MARGIN * t['eager_ms']:
        _slower("decoder", t)
