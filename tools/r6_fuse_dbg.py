"""Round-6 check of Fuse_sft_MLP's native backward chain against float64 with upstream gradients whose ROWS span many orders of
magnitude (what the stage-2 graph feeds it: most points carry almost no compositing weight)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa
from e3dge_amd.local_query import Fuse_sft_MLP
dev = "cuda:0"
torch.manual_seed(0)
m = Fuse_sft_MLP().to(dev)
with torch.no_grad():
    for p in m.parameters():
        p.copy_(torch.randn_like(p) * (0.1 if p.ndim == 1 else 1.0 / p.shape[1] ** 0.5))
N = 6144
x = torch.randn(1, N, 513, device=dev)
for spread in (0.0, 6.0):
    g = torch.randn(1, N, 256, device=dev) * (10.0 ** (-spread * torch.rand(1, N, 1, device=dev)))
    res = {}
    for mode in ("hip", "f64"):
        mm = m if mode == "hip" else Fuse_sft_MLP().double().to(dev)
        if mode == "f64":
            mm.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
            os.environ["E3DGE_FUSE_AUTOGRAD"] = "torch"
        xr = (x if mode == "hip" else x.double()).clone().requires_grad_(True)
        for p in mm.parameters():
            p.grad = None
        y = mm.fuse(xr, xr[..., 257:])
        y.backward(g if mode == "hip" else g.double())
        res[mode] = dict(y=y.detach(), dx=xr.grad, **{n: p.grad for n, p in mm.named_parameters()})
        os.environ.pop("E3DGE_FUSE_AUTOGRAD", None)
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(f"row magnitudes over 10^{spread:g}: " + "  ".join(f"{k} {l2(res['hip'][k], res['f64'][k]):.1e}" for k in res['hip']))

# ---- the texture head's data gradient with the same kind of upstream rows ----
from e3dge_amd.volume_renderer import ResnetBlockFC
from e3dge_amd import synthetic as syn
h = ResnetBlockFC(301, 512).to(dev)
h.load_state_dict({k: v.to(dev) for k, v in syn.stage2_head_state(h.state_dict()).items()})
h64 = ResnetBlockFC(301, 512).double().to(dev)
h64.load_state_dict({k: v.double() for k, v in h.state_dict().items()})
f = torch.randn(N, 301, device=dev) * (0.2 + 3.0 * torch.rand(1, 301, device=dev))
for spread in (0.0, 6.0):
    ga = torch.randn(N, 256, device=dev) * (10.0 ** (-spread * torch.rand(N, 1, device=dev)))
    gb = torch.randn(N, 256, device=dev) * (10.0 ** (-spread * torch.rand(N, 1, device=dev)))
    xr = f.clone().requires_grad_(True)
    a_, b_ = h.tex_modulations(xr)
    (a_ * ga).sum().add((b_ * gb).sum()).backward()
    x64 = f.double().clone().requires_grad_(True)
    o = h64.shortcut(x64) + h64.fc_1(torch.relu(h64.fc_0(torch.relu(x64))))
    ((o[:, :256] * ga.double()).sum() + (o[:, 256:] * gb.double()).sum()).backward()
    d, t = xr.grad.double(), x64.grad
    rows = (d - t).norm(dim=1) / t.norm(dim=1).clamp_min(1e-300)
    print(f"head d feats, row magnitudes over 10^{spread:g}: l2 {float((d - t).norm() / t.norm()):.1e}; per-row relative error: median {float(rows.median()):.1e} max {float(rows.max()):.1e}")
