"""Round-6 debugging aid: the stage-2 graph of tests/test_gpu_stage2.py WITHOUT the decoder, intermediate gradients (d alpha, d beta, d fused)
of the HIP path against float64 autograd of the restatement, row by row -- which link of the chain carries the 1e-4?"""
import os, sys, json
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd import local_query as lq
from e3dge_amd.volume_renderer import VolumeFeatureRenderer, ResnetBlockFC
from conftest import full_state_dict, load_golden
from oracle import local_ref, renderer_ref
DEV = "cuda:0"
gold = load_golden("grads_stage2_16x24")
res, S, size, cm, fh = (int(gold[k]) for k in ("res", "n_samples", "size", "channel_multiplier", "map_hw"))
g, sd = full_state_dict(size=size, cm=cm, res=res, n_samples=S)
inp = syn.stage2_inputs(res, S, size, 256, fh, seed=int(gold["inputs_seed"]))
fuse = lq.Fuse_sft_MLP(257, 256); fsd = syn.stage2_fuse_state(fuse.state_dict()); fuse.load_state_dict(fsd)
wr, wd = syn.synthetic_inputs(1, seed=int(gold["styles_seed"]))
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), out_im_res=res, mode='test')
hk = 'local_feat_to_tex_modulations_linear.'
hsd = syn.stage2_head_state({k.split(hk)[1]: v for k, v in r.state_dict().items() if hk in k})
own = {k: (hsd[k.split(hk)[1]] if hk in k else (v if 'netLocal' in k else sd['renderer.' + k.replace('network.netGlobal.', 'network.')])) for k, v in r.state_dict().items()}
r.load_state_dict(own); r = r.to(DEV); r.requires_grad_(False)
head = r.network.netLocal.local_feat_to_tex_modulations_linear
head.requires_grad_(True); fuse = fuse.to(DEV); fuse.requires_grad_(True)
T = lambda k: torch.from_numpy(gold[k]).to(DEV)
cam = (T('poses'), T('focal'), T('near'), T('far'))
with torch.no_grad():
    p1 = r(*cam, styles=wr.to(DEV))
rs = np.random.RandomState(77)
g_feat = torch.from_numpy((rs.standard_normal((1, 256, res, res)) / 256.0).astype(np.float32))
cap = {}
orig_fuse, orig_tex = lq.Fuse_sft_MLP.fuse, ResnetBlockFC.tex_modulations
def fuse_hook(self, *a, **k):
    cap['enc_in'] = a[0].detach().clone()
    y = orig_fuse(self, *a, **k)
    if y.requires_grad: y.register_hook(lambda gr: cap.__setitem__('d_fused', gr.detach().clone()))
    return y
def tex_hook(self, x, *a, **k):
    al, be = orig_tex(self, x, *a, **k)
    if al.requires_grad:
        al.register_hook(lambda gr: cap.__setitem__('d_alpha', gr.detach().clone())); be.register_hook(lambda gr: cap.__setitem__('d_beta', gr.detach().clone()))
        if x.requires_grad: x.register_hook(lambda gr: cap.__setitem__('d_feats', gr.detach().clone()))
    return al, be
lq.Fuse_sft_MLP.fuse, ResnetBlockFC.tex_modulations = fuse_hook, tex_hook
os.environ["E3DGE_LAZY_TEX"] = "0"
rm, qm = inp['ref_map'].to(DEV).requires_grad_(True), inp['que_map'].to(DEV).requires_grad_(True)
s = wr.to(DEV).clone().requires_grad_(True)
out = r(*cam, styles=s, local_data_batch=dict(feature_maps=dict(ref=rm, que=qm), ref_calibs=T('ref_calibs'), que_calibs=T('que_calibs'),
                                              points=p1['points'], xyz=p1['xyz'], fuse_sft_block=fuse))
((out['features'] * g_feat.to(DEV)).sum() + (out['gen_thumb_imgs'] * inp['g_rgb'].to(DEV)).sum()).backward()
print("captured on the HIP path:", {k: tuple(v.shape) for k, v in cap.items()})
# float64 restatement with the intermediates kept
c = lambda t: t.detach().cpu()
d = lambda t: t.detach().cpu().double()
leaf = lambda t: d(t).clone().requires_grad_(True)
rm64, qm64, w64 = leaf(inp['ref_map']), leaf(inp['que_map']), leaf(wr)
fs = {syn.STAGE2_FUSE_PREFIX + k: d(v) for k, v in fsd.items()}
hs = {syn.STAGE2_HEAD_PREFIX + k: d(v) for k, v in hsd.items()}
f_, _ = local_ref.local_features(fs, syn.STAGE2_FUSE_PREFIX, d(p1['points']), d(p1['xyz']), rm64, qm64, d(T('ref_calibs')), d(T('que_calibs')))
f_.retain_grad()
al, be = renderer_ref.tex_modulations({**sd, **hs}, syn.STAGE2_HEAD_PREFIX, f_, dtype=torch.float64)
al.retain_grad(); be.retain_grad()
o = renderer_ref.render(sd, *[c(t) for t in cam], w64, res=res, n_samples=S, tex=(al, be), dtype=torch.float64)
((o['features'] * g_feat.double()).sum() + (o['gen_thumb_imgs'] * inp['g_rgb'].double()).sum()).backward()
ref = dict(d_alpha=al.grad.reshape(-1, 256), d_beta=be.grad.reshape(-1, 256), d_feats=f_.grad.reshape(-1, 301), d_fused=f_.grad.reshape(-1, 301)[:, :256])
for k, t in ref.items():
    if k not in cap: print(k, "not captured"); continue
    a = cap[k].reshape(t.shape).double().cpu()
    rowe = (a - t).norm(dim=1); rown = t.norm(dim=1)
    big = rown > 1e-3 * rown.max()
    print(f"{k:8s} l2 {float((a - t).norm() / t.norm()):.1e}  rows: {int(big.sum())} of {len(rown)} above 1e-3 of the largest; per-row rel err there: median {float((rowe[big] / rown[big]).median()):.1e} "
          f"max {float((rowe[big] / rown[big]).max()):.1e}; worst rows (index, rel err, norm/max): {[(int(i), float(rowe[i] / rown[i]), float(rown[i] / rown.max())) for i in torch.argsort(rowe, descending=True)[:4]]}")

# ---- the fuse block alone: float64 torch modules on the captured input and the captured upstream gradient ----
lq.Fuse_sft_MLP.fuse = orig_fuse
f64m = lq.Fuse_sft_MLP(257, 256).double().to(DEV)
f64m.load_state_dict({k: v.double() for k, v in fuse.state_dict().items()})
os.environ["E3DGE_FUSE_AUTOGRAD"] = "torch"
xe = cap['enc_in'].double().clone().requires_grad_(True)
y = f64m.fuse(xe, xe[..., 257:])
y.backward(cap['d_fused'].double().reshape(y.shape))
os.environ.pop("E3DGE_FUSE_AUTOGRAD")
l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
print("fuse block alone (same input, same upstream gradient), HIP vs float64:", {n: f"{l2(p.grad, q.grad):.1e}" for (n, p), (_, q) in zip(fuse.named_parameters(), f64m.named_parameters())})
x_ = cap['enc_in'].reshape(-1, 513)
print("input rows: all-zero 2D block", int((x_[:, :256].abs().max(1).values == 0).sum()), " all-zero 3D block", int((x_[:, 257:].abs().max(1).values == 0).sum()), "of", x_.shape[0],
      "; |x| max", float(x_.abs().max()))
# ---- the same block with the round-4 backward (library GEMMs on the saved tensors of the native forward) ----
for bwd in ("torch", "hip"):
    os.environ["E3DGE_FUSE_BWD"] = bwd
    for p_ in fuse.parameters():
        p_.grad = None
    xe32 = cap['enc_in'].clone().requires_grad_(True)
    y32 = fuse.fuse(xe32, xe32[..., 257:])
    y32.backward(cap['d_fused'].reshape(y32.shape))
    print(f"E3DGE_FUSE_BWD={bwd}:", {n: f"{l2(p.grad, q.grad):.1e}" for (n, p), (_, q) in zip(fuse.named_parameters(), f64m.named_parameters())}, "dx", f"{l2(xe32.grad, xe.grad):.1e}")
os.environ.pop("E3DGE_FUSE_BWD")
# ---- are these activation sign flips?  hidden pre-activations of the two SFT branches and of fc_0: native forward vs float64 ----
keep = {}
with torch.no_grad():
    fuse._fuse_native(cap['enc_in'], 1, None, 0, keep=keep)
    x64 = cap['enc_in'].double().reshape(-1, 513)
    enc = f64m.encode_enc
    net64 = enc.fc_0(torch.relu(x64))
    e64 = enc.shortcut(x64) + enc.fc_1(torch.relu(net64))
    s64, t64 = f64m.scale[0](e64), f64m.shift[0](e64)
    for name, mine, ref in (("net (relu)", keep['net'], net64), ("s1 (lrelu, scale branch)", keep['s1'], s64), ("t1 (lrelu, shift branch)", keep['t1'], t64)):
        flip = (mine.reshape(ref.shape) > 0) != (ref > 0)
        print(f"{name}: {int(flip.sum())} of {ref.numel()} signs differ; |float64 value| there: {[float(v) for v in ref[flip].abs()[:6]]}; max |mine - f64| {float((mine.reshape(ref.shape).double() - torch.where(ref > 0, ref, (0.2 if 'lrelu' in name else 1.0) * ref)).abs().max()):.1e}")
