"""Per-phase shader-cycle sums of e3dge_modconv3x3 from a -DE3DGE_MC_TIMING build (E3DGE_LIB_PATH must point at it):
workgroup 0 / thread 0 accumulates s_memtime deltas at the phase boundaries of every step and leaves them in the unused
floats of the amax buffer's first line.   E3DGE_LIB_PATH=.../lib_mctiming.so python tools/modconv_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import _lib, synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import StyledConv  # noqa: E402

dev = "cuda:0"
shapes = [(256, 512, 64, False), (512, 256, 64, True), (256, 256, 128, False), (256, 128, 128, True), (128, 128, 256, False),
          (128, 64, 256, True), (64, 64, 512, False), (64, 32, 512, True), (32, 32, 1024, False)]
names = ["issue(dma+loads)", "mfma", "epilogue", "convert+lds-store", "vmcnt-wait", "barrier"]
for ci, co, res, up in shapes:
    m = StyledConv(ci, co, 3, 512, upsample=up)
    sd = {k: syn.synthetic_tensor('decoder.convs.0.' + k, v.shape, ci) for k, v in m.state_dict().items() if not k.endswith('kernel')}
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    x = torch.randn(1, ci, res, res, device=dev)
    style = torch.randn(1, 512, device=dev)
    noise = None if up else torch.randn(1, 1, res, res, device=dev)
    buf = torch.zeros(_lib.AMAX_FLOATS, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m.conv.forward_fused(x, style, noise=noise, noise_weight=m.noise.weight, bias=m.activate.bias, act=not up, out_amax=buf)
        torch.cuda.synchronize()
    d = buf[1:9].cpu().tolist()
    tot, steps = d[6], max(d[7], 1)
    print(f"{ci:4d}->{co:4d} @{res:5d}{' up' if up else '   '}: steps {steps:.0f}, cycles/step {tot / steps:8.0f} | " +
          ", ".join(f"{n} {v / steps:7.0f}" for n, v in zip(names, d[:6])))
