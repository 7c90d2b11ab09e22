"""The stage-2 training step (que_render_given_ref, project/trainers/E3DGE/e3dge_full_runner.py:185-317) end to end against the
reference's OWN autograd: tests/golden/grads_stage2_16x24.npz was recorded by oracle/gen_golden_stage2.py through the reference's
modules (netLocal.query pieces, Fuse_sft_MLP, PosEncoding, ResnetBlockFC, SirenGenerator.forward_tex, volume_integration, Decoder).

  * CPU: the restatement's fp32 autograd (oracle/training_ref.restated_stage2) reproduces the recording -- the oracle is pinned for
    this graph too.
  * GPU: the same graph on the HIP path -- e3dge_local_query (+ its backward), the Fuse_sft_MLP node, e3dge_tex_modulations_fwd /
    _bwd + e3dge_wgrad, the texture-FiLM render and its backward (8-wave kernels, TEX form), e3dge_dec2_forward / _backward -- for
    d feature maps, every Fuse_sft_MLP and texture-head parameter and d styles."""
import json
import os

import numpy as np
import pytest
import torch

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn
from conftest import GOLDEN, full_state_dict, load_golden, record

FUSE, HEAD = syn.STAGE2_FUSE_PREFIX, syn.STAGE2_HEAD_PREFIX


def _sub(t):             # the sub-sampling of oracle/gen_golden_stage2.py
    if t.numel() <= 20000:
        return t
    return t[:, ::4] if t.ndim == 4 else t[::4]


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _gather(fmap, p1, calibs):
    """(1, N, C) bilinear samples of a feature map at the first pass's points (the product's own query kernel)."""
    from e3dge_amd.local_query import query_feature_map
    pts = p1['points'].reshape(1, -1, 3)
    return query_feature_map(pts, calibs, fmap.detach())[0]


def _vis(p1, calibs):
    from e3dge_amd.local_query import query_feature_map
    B, _, H, W = p1['xyz'].shape
    S = p1['points'].shape[3]
    surf = p1['xyz'].reshape(B, 3, H * W).permute(0, 2, 1)
    vis = query_feature_map(surf, calibs)[1]
    return vis.reshape(B, H * W, 1).expand(B, H * W, S).reshape(B, H * W * S, 1).float()


def _setup():
    from e3dge_amd.local_query import Fuse_sft_MLP
    gold = load_golden("grads_stage2_16x24")
    res, S, size, cm, fh = (int(gold[k]) for k in ("res", "n_samples", "size", "channel_multiplier", "map_hw"))
    g, sd = full_state_dict(size=size, cm=cm, res=res, n_samples=S)
    inp = syn.stage2_inputs(res, S, size, 256, fh, seed=int(gold["inputs_seed"]))
    fuse = Fuse_sft_MLP(257, 256)
    fsd = syn.stage2_fuse_state(fuse.state_dict())
    fuse.load_state_dict(fsd)
    wr, wd = syn.synthetic_inputs(1, seed=int(gold["styles_seed"]))
    return gold, g, sd, inp, fuse, fsd, wr, wd[:, :g.decoder.n_latent].contiguous(), res, S


def test_stage2_restatement_reproduces_the_references_autograd():
    """fp32 autograd of the restatement == the reference's recorded gradients: 0.0 on every tensor in the authoring container (the report
    written beside the fixture, asserted below).  On another host the CPU library's summation order differs and lrelu' of the decoder is
    a step function: 2.4e-4 was read on the GPU box's CPU where the recording itself is 2-5e-3 from float64 -- the bound here is 1e-3."""
    from oracle import renderer_ref, training_ref
    gold, g, sd, inp, fuse, fsd, wr, wd, res, S = _setup()
    T = lambda k: torch.from_numpy(gold[k])
    cam = (T('poses'), T('focal'), T('near'), T('far'))
    with torch.no_grad():
        o1 = renderer_ref.render(sd, *cam, wr, res=res, n_samples=S)
    from e3dge_amd.volume_renderer import ResnetBlockFC
    hsd = syn.stage2_head_state(ResnetBlockFC(301, 512).state_dict())
    loss, img, thumb, grads = training_ref.restated_stage2(sd, fsd, hsd, inp, o1['points'], o1['xyz'], T('ref_calibs'), T('que_calibs'),
                                                           cam, wr, wd, res, S, torch.float32)
    assert abs(loss - float(gold['ref_loss'])) <= 1e-5 * abs(float(gold['ref_loss']))
    assert float((img[:, :, ::8, ::8] - T('ref_img_sub8')).abs().max()) <= 1e-5
    keys = [k[4:] for k in gold.files if k.startswith('ref_d_')]
    assert len(keys) == 3 + 13 + 5, keys
    worst = max(_rel(_sub(grads[k]), gold['ref_' + k]) for k in keys)
    assert worst <= 1e-3, worst
    rep = json.load(open(os.path.join(GOLDEN, "grads_stage2_report.json")))
    assert max(rep['restatement_vs_reference'].values()) <= 1e-6 and rep['tex_effect_on_features'] > 1e-2


@pytest.mark.gpu
def test_stage2_step_against_the_references_own_autograd():
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    DEV = "cuda:0"
    gold, g, sd, inp, fuse, fsd, wr, wd, res, S = _setup()
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), out_im_res=res, mode='test')
    head_keys = [k for k in r.state_dict() if 'local_feat_to_tex_modulations_linear' in k]
    hsd = syn.stage2_head_state({k.split('local_feat_to_tex_modulations_linear.')[1]: r.state_dict()[k] for k in head_keys})
    own = {}
    for k, v in r.state_dict().items():
        if 'local_feat_to_tex_modulations_linear' in k:
            own[k] = hsd[k.split('local_feat_to_tex_modulations_linear.')[1]]
        elif 'netLocal' in k:
            own[k] = v
        else:
            own[k] = sd['renderer.' + k.replace('network.netGlobal.', 'network.')]
    r.load_state_dict(own)
    r = r.to(DEV)
    r.requires_grad_(False)
    head = r.network.netLocal.local_feat_to_tex_modulations_linear
    head.requires_grad_(True)                              # stage 2 trains the texture head and the fuse block, the generator stays frozen
    fuse = fuse.to(DEV)
    fuse.requires_grad_(True)
    dec = g.decoder.to(DEV).eval()
    dec.requires_grad_(False)
    T = lambda k: torch.from_numpy(gold[k]).to(DEV)
    cam = (T('poses'), T('focal'), T('near'), T('far'))
    wr_d, wd_d = wr.to(DEV), wd.to(DEV)
    with torch.no_grad():
        p1 = r(*cam, styles=wr_d)
    rm, qm = inp['ref_map'].to(DEV).requires_grad_(True), inp['que_map'].to(DEV).requires_grad_(True)
    s = wr_d.clone().requires_grad_(True)
    out = r(*cam, styles=s, local_data_batch=dict(feature_maps=dict(ref=rm, que=qm), ref_calibs=T('ref_calibs'), que_calibs=T('que_calibs'),
                                                  points=p1['points'], xyz=p1['xyz'], fuse_sft_block=fuse))
    img, _ = dec(out['features'], [wd_d], input_is_latent=True, noise=[n.to(DEV) for n in inp['noises']])
    loss = (img * inp['g_img'].to(DEV)).sum() + (out['gen_thumb_imgs'] * inp['g_rgb'].to(DEV)).sum()
    loss.backward()
    got = dict(d_ref_map=rm.grad, d_que_map=qm.grad, d_styles=s.grad)
    got.update({'d_fuse.' + n: p.grad for n, p in fuse.named_parameters()})
    got.update({'d_head.' + n: p.grad for n, p in head.named_parameters()})
    errs = dict(loss=abs(float(loss.detach()) - float(gold['ref_loss'])) / abs(float(gold['ref_loss'])),
                img=float((img[:, :, ::8, ::8].cpu() - torch.from_numpy(gold['ref_img_sub8'])).abs().max()),
                thumb=float((out['gen_thumb_imgs'].cpu() - torch.from_numpy(gold['ref_thumb'])).abs().max()))
    keys = [k[4:] for k in gold.files if k.startswith('ref_d_')]
    bad = {}
    for k in keys:
        assert got[k] is not None, k
        e_ref, e_64 = _rel(_sub(got[k]), gold['ref_' + k]), _rel(_sub(got[k]), gold['f64_' + k])
        ref_64 = _rel(torch.from_numpy(gold['ref_' + k]), gold['f64_' + k])              # the reference's own fp32 distance from float64
        sum_rel = abs(float(got[k].double().sum()) - float(gold['sum_' + k])) / float(gold['abs_' + k])
        errs[k] = dict(vs_reference=e_ref, vs_f64=e_64, reference_vs_f64=ref_64, sum_rel=sum_rel)
        # 1e-3 relative on gradients, or 3x the reference's own distance from float64 where that is larger (lrelu' of the
        # decoder is a step function: the recording itself is 2-5e-3 from float64 on this input, oracle/gen_golden_stage2.py)
        tol = max(1e-3, 3 * ref_64)
        if not (e_64 <= tol and e_ref <= tol and sum_rel <= 1e-3):
            bad[k] = errs[k]
    record("stage2_step_vs_reference_16x24", **{k: (v if not isinstance(v, dict) else json.dumps(v)) for k, v in errs.items()})
    assert errs['loss'] <= 1e-4 and errs['img'] <= 1e-4 and errs['thumb'] <= 5e-6, errs
    assert not bad, bad
    # The same graph WITHOUT the decoder (loss on the feature map and the thumbnail): no lrelu' between the loss and the path, so the
    # arithmetic of gather / fuse / head / texture-FiLM render and of their backward kernels is held to fp32 accuracy against float64
    # autograd of the restatement (CPU, a few seconds).
    from oracle import renderer_ref, training_ref
    rs = np.random.RandomState(77)
    g_feat = torch.from_numpy((rs.standard_normal((1, 256, res, res)) / 256.0).astype(np.float32))
    for t in [rm, qm, s] + list(fuse.parameters()) + list(head.parameters()):
        t.grad = None
    out = r(*cam, styles=s, local_data_batch=dict(feature_maps=dict(ref=rm, que=qm), ref_calibs=T('ref_calibs'), que_calibs=T('que_calibs'),
                                                  points=p1['points'], xyz=p1['xyz'], fuse_sft_block=fuse))
    ((out['features'] * g_feat.to(DEV)).sum() + (out['gen_thumb_imgs'] * inp['g_rgb'].to(DEV)).sum()).backward()
    got = dict(d_ref_map=rm.grad, d_que_map=qm.grad, d_styles=s.grad)
    got.update({'d_fuse.' + n: p.grad for n, p in fuse.named_parameters()})
    got.update({'d_head.' + n: p.grad for n, p in head.named_parameters()})
    c = lambda t: t.detach().cpu()
    args = (sd, fsd, hsd, inp, c(p1['points']), c(p1['xyz']), c(T('ref_calibs')), c(T('que_calibs')), tuple(c(t) for t in cam), wr, wd, res, S)
    _, _, _, g64 = training_ref.restated_stage2(*args, torch.float64, g_feat=g_feat)
    _, _, _, g32 = training_ref.restated_stage2(*args, torch.float32, g_feat=g_feat)
    l2 = lambda a, b: float((a.detach().double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
    e2 = {k: dict(max_hip=_rel(got[k], g64[k]), max_oracle_fp32=_rel(g32[k], g64[k]), l2_hip=l2(got[k], g64[k]), l2_oracle_fp32=l2(g32[k], g64[k]))
          for k in g64}
    record("stage2_step_without_decoder_vs_f64", **{k: json.dumps(v) for k, v in e2.items()})
    # relu / lrelu inside Fuse_sft_MLP are step functions too: a pre-activation within the forward's rounding of zero takes the other
    # branch (found in round 6 with tools/r6_stage2_dbg.py: ONE of the 1.57 M hidden activations of the shift branch, |value| 9e-8, and
    # with it 1.2e-4 of relative L2 on every gradient downstream of it -- the library backward on the same saved tensors reads the same).
    # So: the arithmetic is held to fp32 accuracy wherever no such element exists, the loose branch is allowed only when the flipped
    # activations are actually THERE, and they are counted.
    from e3dge_amd.local_query import Fuse_sft_MLP
    f64m = Fuse_sft_MLP(257, 256).double().to(DEV)
    f64m.load_state_dict({k: v.double() for k, v in fuse.state_dict().items()})
    with torch.no_grad():
        enc_in = torch.cat([_gather(qm, p1, T('que_calibs')), _vis(p1, T('ref_calibs')), _gather(rm, p1, T('ref_calibs'))], -1)
        keep = {}
        fuse._fuse_native(enc_in, 1, None, 0, keep=keep)
        x64 = enc_in.double().reshape(-1, 513)
        net64 = f64m.encode_enc.fc_0(torch.relu(x64))
        e64 = f64m.encode_enc.shortcut(x64) + f64m.encode_enc.fc_1(torch.relu(net64))
        flips = sum(int(((keep[k].reshape(r_.shape) > 0) != (r_ > 0)).sum())
                    for k, r_ in (("net", net64), ("s1", f64m.scale[0](e64)), ("t1", f64m.shift[0](e64))))
    tight = {k: v for k, v in e2.items() if not (v['l2_hip'] <= max(5e-5, 2 * v['l2_oracle_fp32']) and v['max_hip'] <= max(2e-4, 4 * v['max_oracle_fp32']))}
    record("stage2_step_without_decoder_sign_flips", flipped_hidden_activations=flips, gradients_beyond_fp32_accuracy=len(tight))
    assert flips <= 8, flips
    if flips == 0:
        assert not tight, tight
    worst = {k: v for k, v in e2.items() if not (v['l2_hip'] <= 5e-4 and v['max_hip'] <= 5e-3)}
    assert not worst, worst
    head_and_styles = {k: v for k, v in e2.items() if k.startswith('d_head') or k == 'd_styles'}          # upstream of every step function of the fuse block
    assert all(v['l2_hip'] <= max(5e-5, 2 * v['l2_oracle_fp32']) for v in head_and_styles.values()), head_and_styles
