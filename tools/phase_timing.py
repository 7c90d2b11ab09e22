"""Reads the per-phase cycle counts recorded by a -DE3DGE_PHASE_TIMING build of the render kernel
(E3DGE_LIB_PATH must point at that variant).  Prints cycles per phase for the sub-tiles of workgroup 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.volume_renderer import VolumeFeatureRenderer

dev = "cuda:0"
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=24), out_im_res=64, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
wr, _ = syn.synthetic_inputs(1, device=dev)
poses, focal, near, far, _ = generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
names = ["(gap)", "geometry+layer0", "layers1-7", "sdf+alpha+scan", "view layer", "rgb+composite+merge"]
with torch.no_grad():
    for it in range(3):
        out = r(poses, focal, near, far, styles=wr)
    torch.cuda.synchronize()
    d = out['dists'].reshape(-1)[:18].cpu().tolist()
    sync = out['dists'].reshape(-1)[18:30].cpu().tolist()
for sub in range(3):
    print(f"sub-tile {sub}: " + ", ".join(f"{names[i]}={d[sub * 6 + i]:.0f}" for i in range(6)))
tot = sum(d)
if r.siren.mfma_mode == "f16x3":      # 8-wave kernel: thread 0's totals over 384 tiles
    print(f"chunk sync totals of wave 0 (cycles over 384 tiles): dma-wait={sync[0]:.0f} barrier={sync[1]:.0f}")
    print("sum", tot, " mfma-only per sub-tile would be", 16 * 24 * 128, "per wave,", 2 * 16 * 24 * 128, "per SIMD (two waves)")
    print(f"workgroup 0, thread 0: prologue {sync[2]:.0f} cycles, sub-tile loop {sync[3]:.0f}, per-ray output stores {sync[4]:.0f}")
else:
    print("chunk_sync totals per wave (cycles over 192 tiles): " + "; ".join(f"w{w}: dma-wait={sync[3*w]:.0f} barrier={sync[3*w+1]:.0f} issue={sync[3*w+2]:.0f}" for w in range(4)))
    print("sum", tot, " mfma-only per sub-tile would be", 8224 * 64)
