"""StyleSDF generator glue on the gfx950 ops -- host-side mirror of project/models/stylesdf_model.py:30-1172.

Kept from the reference: class names, constructor arguments, parameter / buffer names (so the `g_ema` state
dict loads unchanged: `style.{0,1,2}.*`, `renderer.*`, `decoder.style.*`, `decoder.conv1.{conv.weight,
conv.modulation.*, noise.weight, bias, activate.bias}`, `decoder.convs.{i}.*`, `decoder.to_rgb1/to_rgbs.{i}.*`,
`decoder.noises.noise_{i}`; blur kernels are buffers) and the `G_pred_latents.forward` keyword surface that
`trainer.py:881-897` drives.

What runs where: every elementwise / FIR / weight-preparation step is a hand-written HIP kernel
(e3dge_amd.op, e3dge_modconv_weights); the dense 3x3 / 1x1 convolutions and the tiny style GEMVs are library
calls (MIOpen / rocBLAS through torch) -- SURVEY.md 8(d) lists them as "library, not hand-written".
Discriminators, legacy encoders and noise projection onto meshes (:1192-1765, :375-457) are out of scope.
"""
import math
import random

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from .op import FusedLeakyReLU, fused_leaky_relu, noise_bias_act, upfirdn2d
from .volume_renderer import VolumeFeatureRenderer, _opt_get


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


class MappingLinear(nn.Module):
    """Renderer mapping-network layer (reference :40-82): linear, then fused lrelu with scale=1 (:73)."""

    def __init__(self, in_dim, out_dim, bias=True, activation=None, is_last=False):
        super().__init__()
        std = 0.25 if is_last else 1
        self.weight = nn.Parameter(std * nn.init.kaiming_normal_(torch.empty(out_dim, in_dim), a=0.2, mode='fan_in',
                                                                nonlinearity='leaky_relu'))
        lim = np.sqrt(1 / in_dim)
        self.bias = nn.Parameter(torch.empty(out_dim).uniform_(-lim, lim)) if bias else None
        self.activation = activation

    def forward(self, input):
        if self.activation is not None:
            return fused_leaky_relu(F.linear(input, self.weight), self.bias, scale=1)
        return F.linear(input, self.weight, bias=self.bias)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def _fir_pads(taps, factor, conv_kernel=None, mode='up'):
    """The pad arithmetic of the reference's Upsample / Downsample / ModulatedConv2d blurs (:105-110,
    :131-136, :283-299)."""
    if conv_kernel is None:                       # plain Upsample / Downsample
        p = taps - factor
        return ((p + 1) // 2 + factor - 1, p // 2) if mode == 'up' else ((p + 1) // 2, p // 2)
    if mode == 'up':                              # blur after a stride-2 transposed conv
        p = (taps - factor) - (conv_kernel - 1)
        return ((p + 1) // 2 + factor - 1, p // 2 + 1)
    p = (taps - factor) + (conv_kernel - 1)       # blur before a stride-2 conv
    return ((p + 1) // 2, p // 2)


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        self.pad = _fir_pads(self.kernel.shape[0], factor, mode='up')

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel))
        self.pad = _fir_pads(self.kernel.shape[0], factor, mode='down')

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    """Reference :210-249 (weight stored / lr_mul, runtime scale 1/sqrt(in) * lr_mul)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if self.activation:
            return fused_leaky_relu(F.linear(input, self.weight * self.scale), self.bias * self.lr_mul)
        return F.linear(input, self.weight * self.scale, bias=self.bias * self.lr_mul)


class ModulatedConv2d(nn.Module):
    """Reference :263-362.  Weight modulation + demodulation is one HIP launch (e3dge_modconv_weights) that
    writes the per-sample weights directly in the grouped-conv (or transposed-conv) layout."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            self.blur = Blur(blur_kernel, pad=_fir_pads(len(blur_kernel), 2, kernel_size, 'up'), upsample_factor=2)
        if downsample:
            self.blur = Blur(blur_kernel, pad=_fir_pads(len(blur_kernel), 2, kernel_size, 'down'))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def _weights(self, s, transpose):
        """s (B, Ci) -> (B*Co, Ci, k, k) or, transposed, (B*Ci, Co, k, k)."""
        B = s.shape[0]
        Co, Ci, k = self.out_channel, self.in_channel, self.kernel_size
        needs_graph = torch.is_grad_enabled() and (s.requires_grad or self.weight.requires_grad)
        if needs_graph:   # autograd path: same arithmetic as GPU torch ops (training through the decoder)
            w = self.scale * self.weight * s.reshape(B, 1, Ci, 1, 1)
            if self.demodulate:
                w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).reshape(B, Co, 1, 1, 1)
            if transpose:
                return w.transpose(1, 2).reshape(B * Ci, Co, k, k)
            return w.reshape(B * Co, Ci, k, k)
        _lib.require_gpu(s, "style")
        out = torch.empty((B * Ci, Co, k, k) if transpose else (B * Co, Ci, k, k), device=s.device, dtype=torch.float32)
        wt = self.weight.detach().contiguous()
        sc = s.contiguous()
        with torch.cuda.device(s.device):
            rc = _lib.load().e3dge_modconv_weights(_lib.ptr(out), _lib.ptr(wt), _lib.ptr(sc), float(self.scale),
                                                   int(self.demodulate), int(transpose), B, Co, Ci, k * k,
                                                   _lib.stream_of(sc))
        _lib.check(rc, "e3dge_modconv_weights")
        return out

    def forward(self, input, style):
        B, Ci, H, W = input.shape
        s = self.modulation(style)
        if self.upsample:
            w = self._weights(s, transpose=True)
            out = F.conv_transpose2d(input.reshape(1, B * Ci, H, W), w, padding=0, stride=2, groups=B)
            out = out.reshape(B, self.out_channel, out.shape[2], out.shape[3])
            return self.blur(out)
        w = self._weights(s, transpose=False)
        if self.downsample:
            input = self.blur(input)
            H, W = input.shape[2:]
            out = F.conv2d(input.reshape(1, B * Ci, H, W), w, padding=0, stride=2, groups=B)
        else:
            out = F.conv2d(input.reshape(1, B * Ci, H, W), w, padding=self.padding, groups=B)
        return out.reshape(B, self.out_channel, out.shape[2], out.shape[3])


class NoiseInjection(nn.Module):
    """Parameter holder (`weight`, zero-initialised, reference :365-370); StyledConv fuses its arithmetic
    into the activation kernel.  Mesh-projected noise (`project=True`, :423-457) is out of scope."""

    def __init__(self, project=False):
        super().__init__()
        if project:
            raise NotImplementedError("project_noise needs pytorch3d mesh rendering (out of scope)")
        self.project = project
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None, transform=None, mesh_path=None):
        if noise is None:
            B, _, H, W = image.shape
            noise = image.new_empty(B, 1, H, W).normal_()
        return noise_bias_act(image, noise, self.weight, None, negative_slope=1.0, scale=1.0)


class StyledConv(nn.Module):
    """mod-conv -> + noise.weight * noise -> lrelu(x + activate.bias, 0.2) * sqrt(2)  (reference :469-507);
    the last two are ONE pass over the activation (e3dge_noise_bias_act)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 project_noise=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel)
        self.noise = NoiseInjection(project=project_noise)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1, 1))   # unused by forward, as in the reference (:491)
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, transform=None, mesh_path=None):
        out = self.conv(input, style)
        if noise is None:
            B, _, H, W = out.shape
            noise = out.new_empty(B, 1, H, W).normal_()
        return noise_bias_act(out, noise, self.noise.weight, self.activate.bias, self.activate.negative_slope,
                              self.activate.scale)


class ToRGB(nn.Module):
    """1x1 modulated conv without demodulation + bias + FIR-up-sampled skip (reference :510-541)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.upsample = Upsample(blur_kernel) if upsample else upsample
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        out = self.conv(input, style) + self.bias
        if skip is not None:
            if self.upsample:
                skip = self.upsample(skip)
            out = out + skip
        return out


class Decoder(nn.Module):
    """64x64x256 feature map -> size x size RGB (reference :587-797)."""

    def __init__(self, model_opt, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.size = model_opt.size
        self.style_dim = model_opt.style_dim * 2
        in_res = model_opt.renderer_spatial_output_dim
        lr_map = model_opt.lr_mapping
        layers = [PixelNorm(), EqualLinear(self.style_dim // 2, self.style_dim, lr_mul=lr_map, activation="fused_lrelu")]
        layers += [EqualLinear(self.style_dim, self.style_dim, lr_mul=lr_map, activation="fused_lrelu")
                   for _ in range(4)]
        self.style = nn.Sequential(*layers)
        cm = model_opt.channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm,
                         512: 32 * cm, 1024: 16 * cm}
        self.log_size = int(math.log(self.size, 2))
        self.log_in_size = int(math.log(in_res, 2))
        project_noise = _opt_get(model_opt, 'project_noise', False)
        self.conv1 = StyledConv(model_opt.feature_encoder_in_channels, self.channels[in_res], 3, self.style_dim,
                                blur_kernel=blur_kernel, project_noise=project_noise)
        self.to_rgb1 = ToRGB(self.channels[in_res], self.style_dim, upsample=False)
        self.num_layers = (self.log_size - self.log_in_size) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 2 * self.log_in_size + 1) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[in_res]
        for i in range(self.log_in_size + 1, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, self.style_dim, upsample=True,
                                         blur_kernel=blur_kernel, project_noise=project_noise))
            self.convs.append(StyledConv(out_channel, out_channel, 3, self.style_dim, blur_kernel=blur_kernel,
                                         project_noise=project_noise))
            self.to_rgbs.append(ToRGB(out_channel, self.style_dim))
            in_channel = out_channel
        self.n_latent = (self.log_size - self.log_in_size) * 2 + 2

    def mean_latent(self, renderer_latent):
        return self.style(renderer_latent).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def styles_and_noise_forward(self, styles, noise, inject_index=None, truncation=1, truncation_latent=None,
                                 input_is_latent=False, randomize_noise=True):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else \
                [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent[1] + truncation * (s - truncation_latent[1]) for s in styles]
        if len(styles) < 2:
            inject_index = self.n_latent
            latent = styles[0] if styles[0].ndim >= 3 else styles[0].unsqueeze(1).repeat(1, inject_index, 1)
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        return latent, noise

    def forward(self, features, styles, rgbd_in=None, transform=None, return_latents=False, inject_index=None,
                truncation=1, truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True,
                mesh_path=None, conditions=None):
        assert isinstance(styles, list), 'wrap latent code with list'
        latent, noise = self.styles_and_noise_forward(styles, noise, inject_index, truncation, truncation_latent,
                                                      input_is_latent, randomize_noise)
        out = self.conv1(features, latent[:, 0], noise=noise[0])
        skip = self.to_rgb1(out, latent[:, 1], skip=rgbd_in)
        i = 1
        for up_conv, conv, n_up, n_conv, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2],
                                                       self.to_rgbs):
            out = up_conv(out, latent[:, i], noise=n_up)
            out = conv(out, latent[:, i + 1], noise=n_conv)
            skip = to_rgb(out, latent[:, i + 2], skip=skip)
            i += 2
        return skip, (latent if return_latents else None)


class Generator(nn.Module):
    """mapping network + volume renderer + decoder (reference :800-1020)."""

    def __init__(self, model_opt, renderer_opt, blur_kernel=[1, 3, 3, 1], ema=False, full_pipeline=True):
        super().__init__()
        self.size = model_opt.size
        self.style_dim = model_opt.style_dim
        self.num_layers = 1
        self.train_renderer = not model_opt.freeze_renderer
        self.full_pipeline = full_pipeline
        model_opt.feature_encoder_in_channels = _opt_get(renderer_opt, 'width', 256)
        self.is_train = not (ema or model_opt.is_test)
        self.style = nn.Sequential(*[MappingLinear(self.style_dim, self.style_dim, activation="fused_lrelu")
                                     for _ in range(3)])
        # the reference builds the renderer in its default mode='train' and relies on perturb=0 from the option
        # overrides (base_setup.py:53-56); same here.
        self.renderer = VolumeFeatureRenderer(renderer_opt, style_dim=self.style_dim,
                                              out_im_res=model_opt.renderer_spatial_output_dim)
        self.renderer_n_latent = _opt_get(renderer_opt, 'depth', 8) + 1
        if self.full_pipeline:
            self.decoder = Decoder(model_opt)
            self.stylegan_n_latent = 10

    def mean_latent(self, n_latent, device):
        latent_in = torch.randn(n_latent, self.style_dim, device=device)
        renderer_latent = self.style(latent_in)
        renderer_latent_mean = renderer_latent.mean(0, keepdim=True)
        decoder_latent_mean = None
        if self.full_pipeline:
            decoder_latent_mean = self.decoder.mean_latent(renderer_latent)
            self.decoder_latent_mean = decoder_latent_mean.to(device)
        return [renderer_latent_mean, decoder_latent_mean]

    def get_latent(self, input):
        return self.style(input)

    def styles_and_noise_forward(self, styles, inject_index=None, truncation=1, truncation_latent=None,
                                 input_is_latent=False):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if truncation < 1:
            assert isinstance(truncation_latent, list)
            styles = [truncation_latent[0] + truncation * (s - truncation_latent[0]) for s in styles]
        return styles


class G_pred_latents(Generator):
    """The generator entry the runners call (reference :1023-1172, call site trainer.py:881-897):
    `generator([w_renderer, w_decoder], cam_poses, focals, near, far, input_is_latent=True, ...) -> dict`."""

    def forward(self, styles, cam_poses, focals, near=0.88, far=1.12, return_latents=False, inject_index=None,
                truncation=1, truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True,
                return_sdf=False, return_xyz=False, return_eikonal=False, project_noise=False, return_mesh=False,
                mesh_with_shading=True, mesh_path=None, conditions=None, sample_mode=False, geometry_sample=None,
                sample_with_decoder=False, sample_with_renderer=False, return_surface_eikonal=False,
                renderer_only=False, inference_mode=False, sample_without_grad=False, **kwargs):
        if project_noise:
            raise NotImplementedError("project_noise is out of scope")
        if self.full_pipeline:
            assert type(styles) in [list, tuple], 'reformat latent to list/tuple'
            if not input_is_latent:
                encoder_latent, decoder_latent = styles[0], None
            else:
                encoder_latent, decoder_latent = styles
        else:
            decoder_latent = None
            encoder_latent = styles[0]
        renderer_latent = self.styles_and_noise_forward([encoder_latent], inject_index, truncation,
                                                        truncation_latent, input_is_latent)
        render_out = self.renderer(cam_poses, focals, near, far, styles=renderer_latent[0],
                                   return_eikonal=return_eikonal, return_mesh=return_mesh,
                                   mesh_with_shading=mesh_with_shading, sample_mode=sample_mode,
                                   geometry_sample=geometry_sample, return_surface_eikonal=return_surface_eikonal,
                                   sample_without_grad=sample_without_grad, **kwargs)
        render_out['styles'] = renderer_latent[0]
        if renderer_only:
            return render_out
        if (self.full_pipeline or sample_with_decoder) and not sample_with_renderer:
            if decoder_latent is None:
                decoder_latent = renderer_latent
            elif not isinstance(decoder_latent, list):
                decoder_latent = [decoder_latent]
            gen_imgs, decoder_latent = self.decoder(
                render_out['features'], decoder_latent, transform=None, return_latents=return_latents,
                inject_index=inject_index, truncation=truncation, truncation_latent=truncation_latent, noise=noise,
                input_is_latent=input_is_latent, randomize_noise=randomize_noise, mesh_path=mesh_path,
                conditions=conditions)
            render_out['gen_imgs'] = gen_imgs
            render_out['decoder_latent'] = decoder_latent
        return render_out
