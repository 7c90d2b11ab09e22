"""Per-phase cycle counts of render pass #2 (the launch that reads pass #1's layer-7 record) from a -DE3DGE_PHASE_TIMING build
(E3DGE_LIB_PATH must point at that variant): workgroup 0, thread 0, the first three sub-tiles + prologue / loop / output totals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.volume_renderer import VolumeFeatureRenderer, _LazyTex, _fuse_texfilm

dev = "cuda:0"
RES, S = 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), out_im_res=RES, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
wr, _ = syn.synthetic_inputs(1, device=dev)
poses, focal, near, far, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
names = ["(gap)", "geometry+record read", "layers1-7 (skipped)", "sdf+alpha+scan (skipped)", "view layer", "rgb+composite+merge"]
g = torch.Generator().manual_seed(3)
tex = (torch.randn(1, RES, RES, S, 256, generator=g).to(dev) * 0.1, torch.randn(1, RES, RES, S, 256, generator=g).to(dev) * 0.1)
feats = torch.randn(1, RES, RES, S, 301, generator=g).to(dev)
head = r.network.netLocal.local_feat_to_tex_modulations_linear
if _fuse_texfilm():
    tex = _LazyTex(head, feats)          # what the inversion forward runs: head + FiLM launch, then this kernel on the FiLM-ed record
print('texture FiLM:', 'fused into the record (head + FiLM launch)' if _fuse_texfilm() else '(alpha, beta) from HBM')
with torch.no_grad():
    film = r.siren.film_params(wr)
    key = r._reuse_key(wr, focal, poses, near, far)
    for it in range(3):
        r.render_with_film(film, focal, poses, near, far, reuse_key=key)                      # pass #1 (+ record)
        out = r.render_with_film(film, focal, poses, near, far, tex_conditions=tex, reuse_key=key)   # pass #2 on the record
    torch.cuda.synchronize()
    d = out['gen_thumb_imgs'].reshape(-1)[:18].cpu().tolist()          # (the timing build of this launch leaves its counts in rgb[0..22])
    sync = out['gen_thumb_imgs'].reshape(-1)[18:30].cpu().tolist()
for sub in range(3):
    print(f"sub-tile {sub}: " + ", ".join(f"{names[i]}={d[sub * 6 + i]:.0f}" for i in range(6)))
print(f"chunk sync totals of wave 0: dma-wait={sync[0]:.0f} barrier={sync[1]:.0f}")
print(f"workgroup 0, thread 0: prologue {sync[2]:.0f} cycles, sub-tile loop {sync[3]:.0f}, per-ray output stores {sync[4]:.0f}")
if os.environ.get("E3DGE_TIMING_DEBUG"):
    print("rgb", out['gen_thumb_imgs'].reshape(-1)[:24].cpu().tolist())
    print("dists", out['dists'].reshape(-1)[:24].cpu().tolist())
    from e3dge_amd import volume_renderer as _vr
    print("record", {k: (v.shape if hasattr(v, 'shape') else type(v)) for k, v in (_vr._BACKBONE.get(r) or {}).items()})
