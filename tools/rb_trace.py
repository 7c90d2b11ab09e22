"""Timeline of the texture-head kernel for one wave (workgroup 7, wave 0, second sub-tile).
  -DE3DGE_RB_TRACE=1: s_memtime at the four phase boundaries, held in SGPRs -- leaves the k-loops' register allocation alone;
                      use this one for numbers (profiles/r2_rb_trace_phases.txt).
  -DE3DGE_RB_TRACE=2: additionally every k-step and chunk wait of one tile per GEMM phase (-DE3DGE_RB_TRACE_T2=1
                      -DE3DGE_RB_TRACE_T3=5).  The stamp code spills inside the loops: structure only.
tools/build_variant.sh rbtrace -DE3DGE_RB_TRACE=1 ; E3DGE_LIB_PATH=cvpr23-e3dge_amd/lib/variants/lib_rbtrace.so python tools/rb_trace.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import _lib  # noqa: E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.volume_renderer import ResnetBlockFC  # noqa: E402

dev = "cuda:0"
prefix = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
h = ResnetBlockFC(301, 512)
h.load_state_dict({k: syn.synthetic_tensor(prefix + k, v.shape) for k, v in h.state_dict().items()})
h = h.to(dev)
feats = syn.synthetic_local_feats(1, 64, 24, device=dev)
with torch.no_grad():
    for _ in range(5):
        h.tex_modulations(feats)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 640)()
lib.e3dge_debug_rb_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.e3dge_debug_rb_trace(buf) == 0
v = list(buf)
NAMES = {100: "chunk wait: enter", 200: "chunk wait: own DMA landed", 300: "chunk wait: barrier passed", 400: "GEMM done",
         500: "epilogue done", 600: "W_s x done", 700: "W_1 r done", 800: "stores issued", 1000: "sub-tile start",
         2000: "x loaded and split", 3000: "phase 2 done", 4000: "phase 3 done"}
t0, prev = v[1], v[1]
for i in range(320):
    tag, t = v[2 * i], v[2 * i + 1]
    if t == 0:
        break
    name = NAMES.get(tag, f"k-step {tag}")
    print(f"{t - t0:8d}  (+{t - prev:6d})  {name}")
    prev = t
