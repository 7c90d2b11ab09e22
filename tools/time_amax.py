"""e3dge_amax / e3dge_amax_rows on a 100-MB operand: us per launch and TB/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import _lib
dev = "cuda:0"
lib = _lib.load()
x = torch.randn(98304, 256, device=dev); y = torch.randn(98304, 301, device=dev)
am = torch.zeros(_lib.AMAX_FLOATS, device=dev)
def ms(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t1 = ms(lambda: lib.e3dge_amax(_lib.ptr(am), _lib.ptr(x), x.numel(), None))
t2 = ms(lambda: lib.e3dge_amax_rows(_lib.ptr(am), _lib.ptr(y), 98304, 256, 301, None))
ok = abs(float(am.max()) - max(float(x.abs().max()), float(y[:, :256].abs().max()))) == 0.0
print(json.dumps({"amax_us": round(1e3 * t1, 2), "amax_TBps": round(x.numel() * 4 / t1 / 1e9, 2), "amax_rows_us": round(1e3 * t2, 2), "exact": ok}))
