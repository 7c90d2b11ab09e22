"""e3dge_wgrad (kernel + fold) per output size and block shape (E3DGE_WGRAD_SHAPE is read once per process: one process per shape).
   python tools/time_wgrad.py            -> one JSON line: {size: ms} for this process's shape, library matmul beside it"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd.wgrad import wgrad, amax_of

dev, P = "cuda:0", 98304
g = torch.Generator(device=dev).manual_seed(0)
out = {"shape": os.environ.get("E3DGE_WGRAD_SHAPE", "auto")}


def ms(fn, n=20):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)


for m, n, gap in ((256, 256, None), (256, 513, 256), (301, 301, None), (256, 301, None), (512, 301, None)):
    a = torch.randn(P, m, device=dev, generator=g)
    b = torch.randn(P, n, device=dev, generator=g)
    ama, amb = amax_of(a), amax_of(b)
    c = torch.empty(m, n, device=dev)
    out[f"{m}x{n}"] = {"hip": ms(lambda: wgrad(a, b, relu_b=True, amax_a=ama, amax_b=amb, out=c, gap_col=gap)),
                       "hip_colsum": ms(lambda: wgrad(a, b, relu_b=True, amax_a=ama, amax_b=amb, out=c, gap_col=gap, colsum=True)),
                       "library": ms(lambda: torch.mm(a.t(), torch.relu(b), out=c)), "colsum_torch": ms(lambda: a.sum(0))}
print(json.dumps(out))
