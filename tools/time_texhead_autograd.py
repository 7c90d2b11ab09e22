"""Texture head (ResnetBlockFC 301 -> 512) on 98,304 points under autograd: forward + backward with the data gradient from
e3dge_tex_modulations_bwd (round 5) vs round 4's library chain; parameters frozen / trainable.  -> gpurun_out/texhead_autograd.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.volume_renderer import ResnetBlockFC  # noqa: E402

dev = "cuda:0"
PREFIX = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'
h = ResnetBlockFC(301, 512)
h.load_state_dict({k: syn.synthetic_tensor(PREFIX + k, v.shape) for k, v in h.state_dict().items()})
h = h.to(dev)
n = 98304
x = torch.randn(n, 301, device=dev)
ga, gb = torch.randn(n, 256, device=dev), torch.randn(n, 256, device=dev)


def timed(fn, it=20):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:       # bring the clock up (it drops while the host is busy: DESIGN.md 5)
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / it, 4)


def fb():
    x_ = x.clone().requires_grad_(True)
    al, be = h.tex_modulations(x_)
    torch.autograd.grad([al, be], [x_] + ([p for p in h.parameters()] if h.fc_0.weight.requires_grad else []), [ga, gb])


res = {}
if len(sys.argv) > 1 and sys.argv[1] == "--profile-trainable":        # for rocprofv3 --kernel-trace --stats: only the trainable HIP step
    h.requires_grad_(True)
    print(json.dumps({"fwd_bwd_ms_hip_trainable": timed(fb)}))
    sys.exit(0)
for params in (False, True):
    h.requires_grad_(params)
    for be in ("hip", "library"):
        os.environ["E3DGE_TEXHEAD_BWD"] = be
        res[f"fwd_bwd_ms_{be}_{'trainable' if params else 'frozen'}"] = timed(fb)
os.environ.pop("E3DGE_TEXHEAD_BWD")
h.requires_grad_(False)
with torch.no_grad():
    res["forward_ms"] = timed(lambda: h.tex_modulations(x))
    res["backward_kernel_ms"] = timed(lambda: h._launch_bwd(x, ga, gb))
    from e3dge_amd.wgrad import wgrad  # noqa: E402
    res["wgrad_512x301_ms"] = timed(lambda: (wgrad(ga, x), wgrad(gb, x)))
    res["matmul_512x301_ms"] = timed(lambda: (ga.t() @ x, gb.t() @ x))
    res["x_clone_ms"] = timed(lambda: x.clone())
flop = 2 * n * (301 * 301 * 2 + 2 * 512 * 301)
res["backward_algorithmic_gflop"] = round(flop / 1e9, 2)
res["backward_frac_of_f16_third"] = round(flop / (res["backward_kernel_ms"] * 1e-3) / (2.5e15 / 3), 3)
line = json.dumps({"what": "texture head 301 -> 512, 98,304 points", **res})
print(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/texhead_autograd.json", "w").write(line + "\n")
