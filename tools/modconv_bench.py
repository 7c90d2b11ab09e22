"""Times the 1024^2 decoder's 3x3 layers on the fused kernel (and the library path) with HIP events; run under
`rocprofv3 --kernel-trace --stats` for per-kernel durations.   python tools/modconv_bench.py [hip|library|both]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import StyledConv  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = "cuda:0"
shapes = [(256, 512, 64, False), (512, 256, 64, True), (256, 256, 128, False), (256, 128, 128, True), (128, 128, 256, False),
          (128, 64, 256, True), (64, 64, 512, False), (64, 32, 512, True), (32, 32, 1024, False)]
tot = {}
for ci, co, res, up in shapes:
    m = StyledConv(ci, co, 3, 512, upsample=up)
    sd = {k: syn.synthetic_tensor('decoder.convs.0.' + k, v.shape, ci) for k, v in m.state_dict().items() if not k.endswith('kernel')}
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    x = torch.randn(1, ci, res, res, device=dev)
    style = torch.randn(1, 512, device=dev)
    ores = 2 * res if up else res
    noise = torch.randn(1, 1, ores, ores, device=dev)
    line = f"{ci:4d}->{co:4d} @{res:5d}{' up' if up else '   '}"
    for be in (("hip", "library") if which == "both" else (which,)):
        os.environ["E3DGE_MODCONV"] = be
        with torch.no_grad():
            for _ in range(3):
                m(x, style, noise=noise)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                m(x, style, noise=noise)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tot[be] = tot.get(be, 0.0) + ms
        line += f" | {be}: {ms:8.3f} ms {2 * 9 * ci * co * res * res / ms / 1e9:7.1f} TF"
    print(line, flush=True)
print("total:", {k: round(v, 3) for k, v in tot.items()})
