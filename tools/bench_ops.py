"""HBM roofline of the stream ops at the decoder's real sizes (B=1, 1024^2 generator): GB/s = algorithmic bytes /
mean kernel time (HIP events around a replayed HIP graph of 30 launches), against 8 TB/s peak (6.3 TB/s achievable, MI355X_MICROARCH.md).
Prints one JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e3dge_amd  # noqa
from e3dge_amd import op
from e3dge_amd.stylesdf_model import make_kernel

dev = "cuda:0"
k4 = (make_kernel([1, 3, 3, 1]) * 4).to(dev)


def timeit(fn, n=30, warm=5):
    """Mean GPU time of one call.  The n calls are captured into one HIP graph and replayed, so the figure is the kernels'
    own back-to-back time: launched one by one from Python the small cases measure the host's launch rate (~17 us per call)
    instead.  Falls back to plain event timing if the capture fails."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        TIMING.append("hip graph replay")
    except Exception as exc:                                   # noqa: BLE001
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        TIMING.append(f"eager launches ({type(exc).__name__})")
    return e0.elapsed_time(e1) / n * 1e-3


TIMING = []
cases = []
with torch.no_grad():
    for C, L in [(256, 128), (128, 256), (64, 512), (32, 1024)]:
        x = torch.randn(1, C, L + 1, L + 1, device=dev)
        t = timeit(lambda: op.upfirdn2d(x, k4, pad=(1, 1)))
        by = 4 * (x.numel() + C * L * L)
        cases.append(dict(op="upfirdn2d blur (up1,down1,4x4)", shape=[C, L + 1, L + 1], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
        a = torch.randn(1, C, L, L, device=dev); b = torch.randn(C, device=dev)
        nz = torch.randn(1, 1, L, L, device=dev); nw = torch.full((1,), 0.1, device=dev)
        t = timeit(lambda: op.fused_leaky_relu(a, b))
        by = 8 * a.numel()
        cases.append(dict(op="fused_bias_act (lrelu fwd)", shape=[C, L, L], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
        # round 5: the half-precision forms (ABI 11 re-templated both ops; algorithmic bytes = 2 per element and direction)
        xh, ah, bh = x.half(), a.half(), b.half()
        t = timeit(lambda: op.upfirdn2d(xh, k4, pad=(1, 1)))
        by = 2 * (xh.numel() + C * L * L)
        cases.append(dict(op="upfirdn2d blur fp16 (up1,down1,4x4)", shape=[C, L + 1, L + 1], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
        t = timeit(lambda: op.fused_leaky_relu(ah, bh))
        by = 4 * ah.numel()
        cases.append(dict(op="fused_bias_act fp16 (lrelu fwd)", shape=[C, L, L], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
        r_ = torch.randn(1, C, L, L, device=dev)
        t = timeit(lambda: op.fused_bias_act(a, None, r_, 3, 1, 0.2, 2 ** 0.5))
        by = 12 * a.numel()
        cases.append(dict(op="fused_bias_act (lrelu bwd: grad = 1, ref = output)", shape=[C, L, L], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
        t = timeit(lambda: op.noise_bias_act(a, nz, nw, b))
        by = 8 * a.numel() + 4 * nz.numel()
        cases.append(dict(op="noise_bias_act (noise+bias+lrelu)", shape=[C, L, L], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
    for L in (64, 128, 256, 512):
        s = torch.randn(1, 3, L, L, device=dev)
        t = timeit(lambda: op.upfirdn2d(s, k4, up=2, pad=(2, 1)))
        by = 4 * (s.numel() + 3 * 4 * L * L)
        cases.append(dict(op="upfirdn2d skip upsample (up2,4x4)", shape=[3, L, L], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
# round 2: the fused kernels of the decoder path
from e3dge_amd import _lib  # noqa: E402
from e3dge_amd.stylesdf_model import StyledConv, ToRGB  # noqa: E402
lib = _lib.load()
with torch.no_grad():
    for C, L in [(256, 128), (128, 256), (64, 512), (32, 1024)]:
        x = torch.randn(1, C, L + 1, L + 1, device=dev)
        y = torch.empty(1, C, L, L, device=dev)
        nz = torch.randn(1, L * L, device=dev); nw = torch.full((1,), 0.1, device=dev); b = torch.randn(C, device=dev)
        am = torch.zeros(_lib.AMAX_FLOATS, device=dev)
        t = timeit(lambda: lib.e3dge_blur_noise_bias_act(y.data_ptr(), x.data_ptr(), k4.data_ptr(), nz.data_ptr(), nw.data_ptr(), b.data_ptr(),
                                                         0.2, 2 ** 0.5, 1, C, L + 1, L + 1, 1, 1, 1, am.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream))
        by = 4 * (x.numel() + y.numel() + nz.numel())
        cases.append(dict(op="blur+noise+bias+lrelu (+amax), one pass", shape=[C, L + 1, L + 1], bytes=by, us=t * 1e6, GBps=by / t / 1e9))
    for C, L in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
        m = ToRGB(C, 512, upsample=True).to(dev).eval()
        x = torch.randn(1, C, L, L, device=dev); style = torch.randn(1, 512, device=dev)
        skip = torch.randn(1, 3, L // 2, L // 2, device=dev)
        t = timeit(lambda: m(x, style, skip=skip))
        by = 4 * (x.numel() + 3 * L * L + skip.numel())
        cases.append(dict(op="ToRGB fused (1x1 modconv + bias + up-sampled skip; includes the modulation GEMV launch)", shape=[C, L, L],
                          bytes=by, us=t * 1e6, GBps=by / t / 1e9))
for c, how in zip(cases, TIMING):
    c["frac_of_8TBps"] = c["GBps"] / 8000.0
    c["timing"] = how
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items()}))
