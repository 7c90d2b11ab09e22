"""CPU restatement of the two StyleGAN2 custom ops (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows the reference's own PyTorch fallbacks (project/models/op/fused_act.py:106-118,
project/models/op/upfirdn2d.py:157-200) and, for the raw op, the CUDA kernel's act/grad table
(project/models/op/fused_bias_act_kernel.cu:19-49)."""
import torch
from torch.nn import functional as F


def fused_bias_act_ref(x, bias, ref, act, grad, alpha, scale):
    """y = act(x + bias[(i / step_b) % size_b]) * scale, table at fused_bias_act_kernel.cu:35-45.
    bias / ref may be None or empty.  step_b = prod(shape[2:]) (:66-71)."""
    x = x.contiguous()
    v = x
    if bias is not None and bias.numel():
        step_b = 1
        for d in x.shape[2:]:
            step_b *= d
        idx = (torch.arange(x.numel()) // step_b) % bias.numel()
        v = x + bias.reshape(-1)[idx].reshape(x.shape)
    r = ref.reshape(x.shape) if (ref is not None and ref.numel()) else torch.zeros_like(x)
    code = act * 10 + grad
    if code in (12, 32):
        y = torch.zeros_like(v)
    elif code == 30:
        y = torch.where(v > 0, v, v * alpha)
    elif code == 31:
        y = torch.where(r > 0, v, v * alpha)
    else:                      # 10, 11 and the kernel's `default`
        y = v
    return y * scale


def fused_leaky_relu_ref(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """fused_act.py:106-118.  The reference's CPU branch hard-codes slope 0.2 (:112,115) while its CUDA
    branch honours the argument; this restatement honours the argument (every caller passes 0.2)."""
    if bias is not None:
        x = x + bias.reshape(1, bias.shape[0], *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope=negative_slope) * scale


def upfirdn2d_ref(x, kernel, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """upfirdn2d.py:157-200: zero-insert by `up`, pad (negative = crop), correlate with the flipped FIR,
    keep every `down`-th sample.  x (B,C,H,W); up/down (x,y); pad (x0,x1,y0,y1)."""
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    planes = x.reshape(B * C, 1, H, W)
    z = planes.new_zeros(B * C, 1, H * uy, W * ux)
    z[:, :, ::uy, ::ux] = planes                                   # samples followed by (up-1) zeros (:167-169)
    z = F.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0):z.shape[2] - max(-py1, 0), max(-px0, 0):z.shape[3] - max(-px1, 0)]
    y = F.conv2d(z, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(z))        # (dtype AND device: the GPU tests run single layers on the device)
    y = y[:, :, ::dy, ::dx]
    out_h = (H * uy + py0 + py1 - kh) // dy + 1
    out_w = (W * ux + px0 + px1 - kw) // dx + 1
    assert y.shape[2] == out_h and y.shape[3] == out_w, (y.shape, out_h, out_w)
    return y.reshape(B, C, out_h, out_w)


def upfirdn2d_ref_simple(x, kernel, up=1, down=1, pad=(0, 0)):
    """Signature of the public wrapper (upfirdn2d.py:145-154)."""
    return upfirdn2d_ref(x, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
