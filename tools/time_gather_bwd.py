"""e3dge_local_query_bwd at the stage-2 step's geometry: 64x64 rays x 24 samples projected into their own view's map (one pixel per ray)
and into the reference view's (a line of pixels per ray), (1,256,128,128) maps: time per launch and what bounds it."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import _lib, synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.volume_renderer import VolumeFeatureRenderer

dev, RES, S = "cuda:0", 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=RES, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev); r.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
w1, _ = syn.synthetic_inputs(1, seed=1, device=dev)
with torch.no_grad():
    pts = r(p1, f1, n1, fa1, styles=w1)['points'].reshape(1, -1, 3).contiguous()
cq = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev), return_calibs=True)['calibs'][:, :3, :4].contiguous()
cr = generate_camera_params(RES, dev, locations=torch.tensor([[-0.2, 0.05]], device=dev), return_calibs=True)['calibs'][:, :3, :4].contiguous()
N = pts.shape[1]
g = torch.randn(1, N, 513, device=dev)
fm = torch.randn(1, 128, 128, 256, device=dev)
d_fm = torch.zeros_like(fm)
lib = _lib.load()


def ms(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)


def bwd(c, off):
    _lib.check(lib.e3dge_local_query_bwd(_lib.ptr(d_fm), None, _lib.ptr(g), 513, off, _lib.ptr(pts), _lib.ptr(c), _lib.ptr(fm), 1, N, 256, 128, 128,
                                         _lib.stream_of(g)), "bwd")


n_ws = lib.e3dge_local_query_sort_ws_ints(1, N, 128, 128)
ws = torch.empty(n_ws, device=dev, dtype=torch.int32)


def bwd_sorted(c, off):
    _lib.check(lib.e3dge_local_query_bwd_sorted(_lib.ptr(d_fm), None, _lib.ptr(g), 513, off, _lib.ptr(pts), _lib.ptr(c), _lib.ptr(fm), 1, N, 256, 128, 128,
                                                _lib.ptr(ws), n_ws, _lib.stream_of(g)), "bwd_sorted")


out = {"que_view_ms": ms(lambda: bwd(cq, 0)), "ref_view_ms": ms(lambda: bwd(cr, 257)),
       "que_view_sorted_ms": ms(lambda: bwd_sorted(cq, 0)), "ref_view_sorted_ms": ms(lambda: bwd_sorted(cr, 257)), "row_read_floor_ms_at_3p7TBps": round(N * 1024 / 3.7e9, 4)}
# how often the pixel changes along the 32-point runs
with torch.no_grad():
    for name, c in (("que", cq), ("ref", cr)):
        h = torch.cat([pts[0], torch.ones(N, 1, device=dev)], 1) @ c[0].t()
        x, y = h[:, 0] / h[:, 2], -h[:, 1] / h[:, 2]
        ix, iy = torch.floor(((x + 1) * 128 - 1) / 2), torch.floor(((y + 1) * 128 - 1) / 2)
        ch = ((ix[1:] != ix[:-1]) | (iy[1:] != iy[:-1])).float().mean().item()
        inside = ((ix >= -1) & (ix < 128) & (iy >= -1) & (iy < 128)).float().mean().item()
        out[name + "_pixel_changes_per_point"] = round(ch, 3); out[name + "_inside"] = round(inside, 3)
print(json.dumps(out))
