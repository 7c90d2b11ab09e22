"""Fuse_sft_MLP under autograd at the inversion forward's size (98,304 points): forward + backward, the native forward of
_FuseFn against the torch modules (E3DGE_FUSE_AUTOGRAD=torch)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd.local_query import Fuse_sft_MLP

dev = "cuda:0"
torch.manual_seed(0)
m = Fuse_sft_MLP().to(dev)
with torch.no_grad():
    for p in m.parameters():
        p.copy_(torch.randn_like(p) * (0.1 if p.ndim == 1 else 1.0 / p.shape[1] ** 0.5))
N = 64 * 64 * 24
x = torch.randn(1, N, 513, device=dev)
g = torch.randn(1, N, 256, device=dev)


def step(fwd_only):
    xr = x.clone().requires_grad_(True)
    y = m.fuse(xr, xr[..., 257:])
    if not fwd_only:
        for p in m.parameters():
            p.grad = None
        y.backward(g)
    return y


def ms(fn, n=10):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:       # bring the clock up (it drops while the host is busy: DESIGN.md 5)
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
if len(sys.argv) > 1 and sys.argv[1] == "--profile-trainable":        # for rocprofv3 --kernel-trace --stats: only the trainable HIP step
    os.environ["E3DGE_FUSE_AUTOGRAD"] = "hip"
    print(json.dumps({"fwd_bwd_ms_hip_trainable": round(ms(lambda: step(False), 20), 3)}))
    sys.exit(0)
for mode in ("hip", "torch"):
    os.environ["E3DGE_FUSE_AUTOGRAD"] = mode
    res[mode] = {"node": type(step(True).grad_fn).__name__, "fwd_ms": round(ms(lambda: step(True)), 3), "fwd_bwd_ms": round(ms(lambda: step(False)), 3)}
# round 5: parameters frozen (bench.py's shape: the data gradient only), native backward chain vs round 4's library GEMMs
m.requires_grad_(False)
os.environ["E3DGE_FUSE_AUTOGRAD"] = "hip"
for bwd in ("hip", "torch"):
    os.environ["E3DGE_FUSE_BWD"] = bwd
    res["frozen_bwd_" + bwd] = {"fwd_bwd_ms": round(ms(lambda: step(False)), 3)}
os.environ.pop("E3DGE_FUSE_BWD", None)
print(json.dumps({"what": "Fuse_sft_MLP under autograd, 98,304 points (times include the clone of the input)", **res}))
