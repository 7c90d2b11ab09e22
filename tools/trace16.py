"""Timeline of one GEMM tile (hidden layer 3, tile 6, workgroup 7) of the forward kernel for two waves that share a SIMD
(waves 0 and 4): s_memtime at the start of every k-step, after its three MFMAs were issued, and after the epilogue slice.
Needs a build with -DE3DGE_16_TRACE (tools/build_variant.sh trace16 -DE3DGE_16_TRACE; E3DGE_LIB_PATH=...)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import _lib  # noqa: E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.volume_renderer import VolumeFeatureRenderer  # noqa: E402

dev, res, S = "cuda:0", 64, 24
r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S), out_im_res=res, mode='test')
syn.load_synthetic(r, prefix='renderer.')
r = r.to(dev)
wr, _ = syn.synthetic_inputs(1, seed=7, device=dev)
poses, focal, near, far, _ = generate_camera_params(res, dev, batch=1)
with torch.no_grad():
    for _ in range(3):
        r(poses, focal, near, far, styles=wr)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 48)()
lib.e3dge_debug_trace16.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.e3dge_debug_trace16(buf) == 0
v = list(buf)
t0 = min(x for x in v if x)
for w in range(2):
    print(f"wave {4 * w}:")
    for g in range(8):
        a, b, c = (v[w * 24 + 3 * g + i] - t0 for i in range(3))
        print(f"   k-step {g}: start {a:6d}   mfma issued {b:6d} (+{b - a:4d})   epilogue done {c:6d} (+{c - b:4d})")
