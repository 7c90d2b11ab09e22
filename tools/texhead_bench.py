"""Time e3dge_tex_modulations_fwd at the C2 size (98,304 points x 301 features).  python tools/texhead_bench.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.volume_renderer import ResnetBlockFC  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
prefix = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
h = ResnetBlockFC(301, 512)
h.load_state_dict({k: syn.synthetic_tensor(prefix + k, v.shape) for k, v in h.state_dict().items()})
h = h.to(dev)
feats = syn.synthetic_local_feats(1, 64, 24, device=dev)
with torch.no_grad():
    for _ in range(3):
        h.tex_modulations(feats)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        h.tex_modulations(feats)
    b.record()
    torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
flops = 2 * (301 * 301 + 2 * 301 * 512) * 64 * 64 * 24
print(f"tex head, 98,304 points x 301: {ms:.3f} ms  ({flops / ms / 1e9:.1f} algorithmic TFLOP/s)")
