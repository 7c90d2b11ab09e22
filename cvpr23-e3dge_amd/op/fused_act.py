"""fused_leaky_relu / FusedLeakyReLU on the HIP kernel e3dge_fused_bias_act.

Public surface = project/models/op/fused_act.py (`fused_leaky_relu(input, bias=None, negative_slope=0.2,
scale=2**0.5)`, `FusedLeakyReLU(channel, bias=True, ...)` with parameter `.bias`); `fused_bias_act` keeps
the argument list of the reference's pybind entry (fused_bias_act.cpp:11-20).

Autograd: y = scale * lrelu(x + b).  dy/dx is the mask m = (y > 0 ? 1 : slope) * scale, which the kernel
evaluates from the saved OUTPUT (act=3, grad=1; fused_bias_act_kernel.cu:42).  `_MaskedScale` applies that
mask to any tensor and is linear in it, so its own backward is `_MaskedScale` again -- any order of
differentiation w.r.t. the input is covered (the reference reaches second order, fused_act.py:19-52).

CPU tensors take a plain-PyTorch branch written here (the reference has one too, fused_act.py:107-118) so that host-side
code -- building modules, shape checks, CPU unit tests of callers -- behaves as with the reference; it is ordinary torch
autograd, not a kernel, and nothing on the GPU path ever routes through it.  Unlike the reference's CPU branch, which
hard-codes the slope 0.2 (:112,115), it honours `negative_slope` like the GPU kernel does (identical at the default)."""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """y = act(x + bias[(i / prod(shape[2:])) % len(bias)]) * scale (see include/e3dge_hip.h).
    `bias` / `refer` may be None or empty tensors, as the reference passes `empty`."""
    _lib.require_gpu(input, "input", half_ok=True)
    x = input.contiguous()
    half = x.dtype == torch.float16          # the reference dispatches on the input's type (fused_bias_act_kernel.cu:79)
    b = bias.contiguous().to(x.dtype) if bias is not None and bias.numel() else None
    r = refer.contiguous().to(x.dtype) if refer is not None and refer.numel() else None
    if b is not None:
        _lib.require_gpu(b, "bias", half_ok=True)
    if r is not None:
        _lib.require_gpu(r, "refer", half_ok=True)
        if r.numel() != x.numel():
            raise RuntimeError("refer must have as many elements as input")
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    y = torch.empty_like(x)
    lib = _lib.load()
    fn = lib.e3dge_fused_bias_act_f16 if half else (lib.e3dge_fused_bias_act_f64 if x.dtype == torch.float64 else lib.e3dge_fused_bias_act)
    with torch.cuda.device(x.device):
        rc = fn(_lib.ptr(y), _lib.ptr(x), _lib.ptr(b), _lib.ptr(r), int(act), int(grad), float(alpha), float(scale),
                x.numel(), step_b, 0 if b is None else b.numel(), _lib.stream_of(x))
    _lib.check(rc, "e3dge_fused_bias_act")
    return y


def _channel_sum(t):
    return t.sum([0] + list(range(2, t.ndim)))


class _MaskedScale(Function):
    """g -> g * (out > 0 ? 1 : slope) * scale, with `out` the saved forward output (treated as constant)."""

    @staticmethod
    def forward(ctx, g, out, slope, scale):
        ctx.save_for_backward(out)
        ctx.slope, ctx.scale = slope, scale
        return fused_bias_act(g, None, out, 3, 1, slope, scale)

    @staticmethod
    def backward(ctx, gg):
        out, = ctx.saved_tensors
        return _MaskedScale.apply(gg, out, ctx.slope, ctx.scale), None, None, None


class _BiasLrelu(Function):
    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        y = fused_bias_act(x, bias, None, 3, 0, slope, scale)
        ctx.save_for_backward(y)
        ctx.slope, ctx.scale, ctx.has_bias = slope, scale, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gx = _MaskedScale.apply(gy, y, ctx.slope, ctx.scale)
        gb = _channel_sum(gx) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return gx, gb, None, None


def _fused_leaky_relu_cpu(input, bias, negative_slope, scale):
    if bias is not None:
        input = input + bias.reshape(1, bias.shape[0], *([1] * (input.ndim - 2)))
    return torch.nn.functional.leaky_relu(input, negative_slope=negative_slope) * scale


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if isinstance(input, torch.Tensor) and input.device.type == "cpu":
        return _fused_leaky_relu_cpu(input, bias, negative_slope, scale)
    _lib.require_gpu(input, "input", half_ok=True)
    return _BiasLrelu.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """State-dict compatible with the reference module (fused_act.py:87-103): one parameter, `bias`."""

    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


class _NoiseBiasLrelu(Function):
    """StyledConv's tail (NoiseInjection, stylesdf_model.py:459-466, then FusedLeakyReLU) in one HBM pass."""

    @staticmethod
    def forward(ctx, x, noise, noise_weight, bias, slope, scale):
        B, C = x.shape[0], x.shape[1]
        xc = x.contiguous()
        nz = None if noise is None else noise.contiguous()
        y = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            rc = _lib.load().e3dge_noise_bias_act(
                _lib.ptr(y), _lib.ptr(xc), _lib.ptr(nz), _lib.ptr(noise_weight), _lib.ptr(bias), float(slope),
                float(scale), B, C, xc.numel() // max(B * C, 1), 0 if nz is None else nz.shape[0],
                _lib.stream_of(x))
        _lib.check(rc, "e3dge_noise_bias_act")
        ctx.save_for_backward(y, nz if nz is not None else y.new_empty(0))
        ctx.slope, ctx.scale = slope, scale
        return y

    @staticmethod
    def backward(ctx, gy):
        y, nz = ctx.saved_tensors
        gx = _MaskedScale.apply(gy, y, ctx.slope, ctx.scale)
        g_w = g_b = None
        if nz.numel() and ctx.needs_input_grad[2]:
            g_w = (gx.sum(1, keepdim=True) * nz.reshape(nz.shape[0], 1, *gx.shape[2:])).sum().reshape(1)
        if ctx.needs_input_grad[3]:
            g_b = _channel_sum(gx)
        return gx, None, g_w, g_b, None, None


def noise_bias_act(x, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """lrelu(x + noise_weight * noise + bias[None,:,None,None], slope) * scale for x (B,C,H,W) and noise
    (1|B,1,H,W) or None."""
    if noise is not None and noise_weight is None:
        raise RuntimeError("noise given without noise_weight")
    if isinstance(x, torch.Tensor) and x.device.type == "cpu":
        if noise is not None:
            x = x + noise_weight * noise
        return _fused_leaky_relu_cpu(x, bias, negative_slope, scale)
    _lib.require_gpu(x, "x")
    if noise is not None:
        _lib.require_gpu(noise, "noise")
    return _NoiseBiasLrelu.apply(x, noise, noise_weight, bias, negative_slope, scale)
