"""CPU restatement of the StyleGAN2 up-sampler of StyleSDF (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional PyTorch over a state dict with the reference's key names.
Reference: project/models/stylesdf_model.py -- line numbers cited per function."""
import math

import torch
from torch.nn import functional as F

from .ops_ref import fused_leaky_relu_ref, upfirdn2d_ref_simple


def _w(sd, key, dtype):
    return sd[key].to(dtype)


def fir(taps=(1, 3, 3, 1), gain=1.0, dtype=torch.float32):
    """make_kernel :85-93 (outer product, normalised), times `gain` (Blur :155-156, Upsample :102)."""
    k = torch.tensor(taps, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return (k / k.sum() * gain).to(dtype)


def equal_linear(sd, prefix, x, lr_mul=1.0, activation=False):
    """EqualLinear.forward :234-244."""
    w = _w(sd, prefix + 'weight', x.dtype)
    b = _w(sd, prefix + 'bias', x.dtype)
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu_ref(F.linear(x, w * scale), b * lr_mul)
    return F.linear(x, w * scale, bias=b * lr_mul)


def mapping_linear(sd, prefix, x):
    """MappingLinear.forward :70-77 with activation: linear without bias, then fused lrelu(scale=1)."""
    return fused_leaky_relu_ref(F.linear(x, _w(sd, prefix + 'weight', x.dtype)), _w(sd, prefix + 'bias', x.dtype), scale=1)


def renderer_mapping(sd, z, prefix='style.'):
    """Generator.style :822-830: three MappingLinear layers."""
    for i in range(3):
        z = mapping_linear(sd, f'{prefix}{i}.', z)
    return z


def decoder_mapping(sd, w, prefix='decoder.style.', lr_mul=0.01):
    """Decoder.style :596-611: PixelNorm + 5 EqualLinear(fused_lrelu)."""
    x = w * torch.rsqrt(torch.mean(w ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, 6):
        x = equal_linear(sd, f'{prefix}{i}.', x, lr_mul, activation=True)
    return x


def modulated_weights(weight, s, demodulate=True):
    """The per-sample kernels of ModulatedConv2d.forward: weight (1, Co, Ci, k, k), modulation s (B, Ci) ->
    (B, Co, Ci, k, k) = scale * weight * s (:321) [* rsqrt(sum over (Ci, k, k) of its square + 1e-8) (:325-326)]."""
    _, Co, Ci, k, _ = weight.shape
    B = s.shape[0]
    w = (1 / math.sqrt(Ci * k * k)) * weight * s.reshape(B, 1, Ci, 1, 1)
    if demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).reshape(B, Co, 1, 1, 1)
    return w


def modulated_conv(sd, prefix, x, style, demodulate=True, upsample=False):
    """ModulatedConv2d.forward :317-362 (no downsample branch on this path)."""
    dt = x.dtype
    weight = _w(sd, prefix + 'weight', dt)                          # (1, Co, Ci, k, k)
    _, Co, Ci, k, _ = weight.shape
    B, _, H, W = x.shape
    s = equal_linear(sd, prefix + 'modulation.', style)             # bias_init 1 lives in the weights
    w = modulated_weights(weight, s, demodulate)
    if upsample:
        wt = w.transpose(1, 2).reshape(B * Ci, Co, k, k)            # :333-338
        out = F.conv_transpose2d(x.reshape(1, B * Ci, H, W), wt, padding=0, stride=2, groups=B)
        out = out.reshape(B, Co, out.shape[2], out.shape[3])
        p = (4 - 2) - (k - 1)                                       # :285-287
        return upfirdn2d_ref_simple(out, fir(gain=4.0, dtype=dt), pad=((p + 1) // 2 + 1, p // 2 + 1))
    out = F.conv2d(x.reshape(1, B * Ci, H, W), w.reshape(B * Co, Ci, k, k), padding=k // 2, groups=B)
    return out.reshape(B, Co, out.shape[2], out.shape[3])


def styled_conv(sd, prefix, x, style, noise, upsample=False):
    """StyledConv.forward :494-507: mod-conv, + noise.weight * noise (:466), lrelu(x + activate.bias) * sqrt 2."""
    out = modulated_conv(sd, prefix + 'conv.', x, style, True, upsample)
    out = out + _w(sd, prefix + 'noise.weight', x.dtype) * noise.to(x.dtype)
    return fused_leaky_relu_ref(out, _w(sd, prefix + 'activate.bias', x.dtype))


def to_rgb(sd, prefix, x, style, skip=None, upsample=True):
    """ToRGB.forward :531-541."""
    out = modulated_conv(sd, prefix + 'conv.', x, style, demodulate=False) + _w(sd, prefix + 'bias', x.dtype)
    if skip is not None:
        if upsample:
            skip = upfirdn2d_ref_simple(skip, fir(gain=4.0, dtype=x.dtype), up=2, pad=(2, 1))   # Upsample :96-119
        out = out + skip
    return out


def decoder_forward(sd, features, latent, noises=None, prefix='decoder.', dtype=torch.float32):
    """Decoder.forward :742-797 with input_is_latent=True, randomize_noise=False (noise buffers) unless
    `noises` is given.  features (B,256,r,r), latent (B,n_latent,512)."""
    features, latent = features.to(dtype), latent.to(dtype)
    n_up = 0
    while f'{prefix}convs.{2 * n_up}.conv.weight' in sd:
        n_up += 1
    if noises is None:
        noises = [sd[f'{prefix}noises.noise_{i}'] for i in range(2 * n_up + 1)]
    out = styled_conv(sd, prefix + 'conv1.', features, latent[:, 0], noises[0])
    skip = to_rgb(sd, prefix + 'to_rgb1.', out, latent[:, 1], None, upsample=False)
    i = 1
    for u in range(n_up):                                           # :773-792
        out = styled_conv(sd, f'{prefix}convs.{2 * u}.', out, latent[:, i], noises[2 * u + 1], upsample=True)
        out = styled_conv(sd, f'{prefix}convs.{2 * u + 1}.', out, latent[:, i + 1], noises[2 * u + 2])
        skip = to_rgb(sd, f'{prefix}to_rgbs.{u}.', out, latent[:, i + 2], skip)
        i += 2
    return skip
