"""Deterministic synthetic weights, options and inputs (no checkpoints or datasets are reachable offline).

Every tensor is drawn from a legacy numpy RandomState seeded by the CRC32 of its state-dict KEY, so the same
key gets the same values whichever module (ours, the oracle, or the reference imported by the fixture
generator) the state dict is loaded into, on any machine.  Ranges follow the reference initialisers
(volume_renderer.py:53-71, 91-114; stylesdf_model.py:54-64, 179-187, 221-224, 305-308) except where the
reference initialises to zero and a zero would leave a code path untested (noise weights, biases)."""
import math
import zlib

import numpy as np
import torch


class AttrDict(dict):
    """Minimal stand-in for the Munch objects the reference passes around (attribute + key access)."""
    def __getattr__(self, k):
        if k.startswith('__'):                  # copy / pickle probe for dunder hooks: a missing one must raise, not be None
            raise AttributeError(k)
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return AttrDict(dict.copy(self))


def rendering_opt(N_samples=24, **over):
    """The rendering options that reach the renderer after base_setup.py:53-56 / train_setup.py:53-56."""
    o = AttrDict(perturb=0, no_offset_sampling=False, N_samples=N_samples, raw_noise_std=0., return_xyz=True,
                 return_sdf=True, static_viewdirs=True, no_z_normalize=False, spatial_super_sampling_factor=1,
                 force_background=True, no_sdf=False, add_fg_mask=False, width=256, depth=8,
                 camera=AttrDict(dist_radius=0.12, fov=6, azim=0.3, elev=0.15, uniform=False),
                 enable_local_model=False, return_feats=False, return_feats_layers=[1, 3, 5, 7],
                 local_modulation_layer_in_backbone=False, local_modulation_layer=False,
                 use_integrated_surface_normal=False, sample_near_surface=False, sample_uniform_grid=False,
                 L_pred_tex_modulations=False, residual_local_feats_dim=301)
    o.update(over)
    return o


def model_opt(size=1024, channel_multiplier=2, renderer_spatial_output_dim=64, **over):
    o = AttrDict(size=size, style_dim=256, channel_multiplier=channel_multiplier, lr_mapping=0.01,
                 renderer_spatial_output_dim=renderer_spatial_output_dim, project_noise=False,
                 freeze_renderer=True, is_test=True)
    o.update(over)
    return o


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def _kaiming_std(fan_in, a=0.2):
    return math.sqrt(2.0 / (1 + a * a)) / math.sqrt(fan_in)


def synthetic_tensor(key, shape, seed=0):
    """Value for one state-dict entry, chosen from its name."""
    rs = _rs(key, seed)
    shape = tuple(shape)
    leaf = key.split('.')[-1]
    u = lambda lim: rs.uniform(-lim, lim, size=shape)
    if key.endswith('sigmoid_beta'):
        return torch.full(shape, 0.1, dtype=torch.float32)
    in_renderer = '.pts_linears.' in key or '.views_linears.' in key or '.rgb_linear.' in key or '.sigma_linear.' in key
    if in_renderer:
        if '.gamma.' in key or '.beta.' in key:                      # LinearLayer(style -> W): kaiming * 0.25
            v = 0.25 * _kaiming_std(256) * rs.standard_normal(shape) if leaf == 'weight' else u(math.sqrt(1 / 256))
        elif '.pts_linears.0.' in key:
            v = u(1 / 3) if leaf == 'weight' else u(math.sqrt(1 / 3))
        elif '.rgb_linear.' in key or '.sigma_linear.' in key:
            v = u(math.sqrt(6 / 256) / 25) if leaf == 'weight' else u(math.sqrt(1 / 256))
        else:                                                        # FiLMSiren 256(+3) -> 256
            fan_in = shape[1] if leaf == 'weight' else (259 if '.views_linears.' in key else 256)
            v = u(math.sqrt(6 / fan_in) / 25) if leaf == 'weight' else u(math.sqrt(1 / fan_in))
        return torch.from_numpy(np.asarray(v, dtype=np.float32))
    if '.local_feat_to_tex_modulations_linear.' in key:              # texture head: the reference zero-inits it (untestable)
        v = math.sqrt(2.0 / shape[-1]) * rs.standard_normal(shape) * (0.5 if '.fc_1.' in key else 1.0) \
            if leaf == 'weight' else 0.05 * rs.standard_normal(shape)
        return torch.from_numpy(np.asarray(v, dtype=np.float32))
    if key.startswith('style.') or ('.style.' not in key and key.split('.')[0] == 'style'):
        v = _kaiming_std(shape[-1]) * rs.standard_normal(shape) if leaf == 'weight' else u(math.sqrt(1 / 256))
        return torch.from_numpy(np.asarray(v, dtype=np.float32))
    # ---- decoder ----
    if '.noises.' in key:
        v = rs.standard_normal(shape)
    elif key.endswith('noise.weight'):
        v = 0.1 + 0.05 * rs.standard_normal(shape)                   # reference init is 0 (:370): untestable
    elif key.endswith('modulation.bias'):
        v = 1.0 + 0.05 * rs.standard_normal(shape)                   # bias_init = 1
    elif key.endswith('activate.bias') or (leaf == 'bias' and len(shape) == 4):
        v = 0.1 * rs.standard_normal(shape)                          # reference init is 0
    elif '.style.' in key and leaf == 'weight':
        v = rs.standard_normal(shape) / 0.01                         # EqualLinear(lr_mul=0.01): randn / lr_mul
    elif leaf == 'weight':
        v = rs.standard_normal(shape)
    elif leaf == 'kernel':
        raise KeyError("blur kernels are deterministic buffers, not synthetic")
    else:
        v = 0.05 * rs.standard_normal(shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32))


def synthetic_state_dict(module, seed=0, prefix=''):
    """Deterministic values for every parameter / persistent buffer of `module` (blur kernels untouched).
    `prefix` is the module's path inside the full generator (e.g. 'renderer.' for a stand-alone
    VolumeFeatureRenderer) so that it receives the same values it would get as a sub-module."""
    out = {}
    for k, v in module.state_dict().items():
        if k.endswith('.kernel'):
            continue
        out[k] = synthetic_tensor(prefix + k, v.shape, seed)
    return out


def load_synthetic(module, seed=0, prefix=''):
    sd = synthetic_state_dict(module, seed, prefix)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith('.kernel') for m in missing), missing
    return module


STRESS_VARIANTS = {            # name: (SIREN hidden-weight factor, gamma-mapping weight factor, std of the W+ codes)
    "wide": (1.0, 1.0, 1.0),    # unit-variance styles: FiLM frequencies 13.5 .. 48 instead of 27.7 .. 32.3
    "s2": (2.0, 3.0, 0.3),      # per-layer gain 2: fp32 rounding amplified ~300x over the init-range fixtures, still well defined
    "x4": (4.0, 3.0, 1.0),      # per-layer gain 4 and above: the network is chaotic -- the reference's own fp32 output is O(1) away
    "x32": (32.0, 3.0, 1.0),    # from float64; kept because real checkpoints may sit anywhere
}


def stress_state_dict(sd, variant, seed=99):
    """Trained-like magnitudes on top of the init-range synthetic weights (tests/golden/stress_*.npz): the default
    split-f16 contraction stores 128 w as f16 hi + lo and FiLM frequencies gamma ~ 30 amplify every rounding, so the parity fixtures
    must not live at |w| <= 0.006 only.  SIREN hidden weights (pts_linears.1-7, views_linears) and the gamma-mapping weights are
    scaled per STRESS_VARIANTS; decoder 3x3 weights become heavy-tailed (student-t, 3 degrees of freedom, clipped at 40), ToRGB
    weights x2, noise weights 0.5.  Returns a new dict; values are deterministic."""
    k_hidden, k_gamma, _ = STRESS_VARIANTS[variant]
    out = {}
    for k, v in sd.items():
        v = v.clone()
        leaf = k.split('.')[-1]
        if ('.pts_linears.' in k or '.views_linears.' in k) and leaf == 'weight':
            if '.gamma.' in k:
                v = v * k_gamma
            elif '.beta.' not in k and '.pts_linears.0.' not in k:
                v = v * k_hidden
        elif '.convs.' in k or '.conv1.' in k:
            if k.endswith('conv.weight'):
                rs = _rs(k, seed)
                t = rs.standard_t(3, size=tuple(v.shape)).astype(np.float32)
                v = torch.from_numpy(np.clip(t, -40.0, 40.0))
            elif k.endswith('noise.weight'):
                v = torch.full_like(v, 0.5)
        elif '.to_rgb' in k and k.endswith('conv.weight'):
            v = v * 2.0
        out[k] = v
    return out


def stress_inputs(variant, batch=1, seed=21, device="cpu"):
    """W+ codes with the variant's standard deviation for the renderer, unit variance for the decoder (the default
    synthetic_inputs are 0.1 * N(0,1))."""
    w_r = STRESS_VARIANTS[variant][2] * np.random.RandomState(seed).standard_normal((batch, 9, 256))
    w_d = np.random.RandomState(seed + 1).standard_normal((batch, 10, 512))
    return (torch.from_numpy(w_r.astype(np.float32)).to(device), torch.from_numpy(w_d.astype(np.float32)).to(device))


def synthetic_inputs(batch=1, seed=1, device="cpu"):
    """W+ codes for the renderer (B,9,256) and the decoder (B,10,512): 0.1 * N(0,1)."""
    w_r = 0.1 * np.random.RandomState(seed).standard_normal((batch, 9, 256))
    w_d = 0.1 * np.random.RandomState(seed + 1).standard_normal((batch, 10, 512))
    return (torch.from_numpy(w_r.astype(np.float32)).to(device), torch.from_numpy(w_d.astype(np.float32)).to(device))


def decoder_grad_inputs(batch, size, in_res, seed=11, device="cpu", channels=256):
    """Inputs of the decoder-gradient fixtures (oracle/gen_golden_decoder_grads.py): a feature map 0.5 * N(0,1) (B, 256, r, r), one
    per-sample noise image per StyledConv (conv1, then two per level), and the upstream gradient N(0,1) / size of the image."""
    rs = np.random.RandomState(seed)
    feats = torch.from_numpy((0.5 * rs.standard_normal((batch, channels, in_res, in_res))).astype(np.float32))
    noises, r = [], in_res
    while r <= size:
        for _ in range(1 if r == in_res else 2):
            noises.append(torch.from_numpy(rs.standard_normal((batch, 1, r, r)).astype(np.float32)))
        r *= 2
    gy = torch.from_numpy(rs.standard_normal((batch, 3, size, size)).astype(np.float32)) / float(size)
    return feats.to(device), [n.to(device) for n in noises], gy.to(device)


def synthetic_tex_conditions(batch, res, n_samples, seed=5, device="cpu"):
    """Per-point texture FiLM (alpha, beta), each (B,H,W,S,256): what the local branch would hand over."""
    rs = np.random.RandomState(seed)
    shape = (batch, res, res, n_samples, 256)
    a = 0.1 * rs.standard_normal(shape).astype(np.float32)
    b = 0.05 * rs.standard_normal(shape).astype(np.float32)
    return torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)


def synthetic_local_feats(batch, res, n_samples, cin=301, seed=5, device='cpu'):
    """(B, res, res, n_samples, cin) stand-in for the PIFu branch's per-point local features (feature-map samples and a
    positional encoding in the reference): mixed magnitudes, some channels large."""
    rs = np.random.RandomState(seed)
    f = rs.standard_normal((batch, res, res, n_samples, cin)).astype(np.float32)
    f *= (0.2 + 3.0 * rs.uniform(size=(1, 1, 1, 1, cin))).astype(np.float32)
    return torch.from_numpy(f).to(device)


# ---- stage-2 step fixture (oracle/gen_golden_stage2.py, tests/test_gpu_stage2.py, bench.py): inputs and the two trainable modules ----
STAGE2_FUSE_PREFIX = 'Fuse_sft_block.'
STAGE2_HEAD_PREFIX = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'


def stage2_inputs(res, n_samples, size, channels=256, map_hw=16, seed=31, device="cpu"):
    """Feature maps of the reference / query view (B = 1; stand-ins for the hourglass filters' outputs), the decoder's per-layer noise
    images and the two fixed upstream gradients of L = <g_img, image> + <g_rgb, thumbnail>."""
    rs = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)
    out = dict(ref_map=t(rs.standard_normal((1, channels, map_hw, map_hw))), que_map=t(rs.standard_normal((1, channels, map_hw, map_hw))))
    noises, r = [], res
    while r <= size:
        for _ in range(1 if r == res else 2):
            noises.append(t(rs.standard_normal((1, 1, r, r))))
        r *= 2
    out['noises'] = noises
    out['g_img'] = t(rs.standard_normal((1, 3, size, size)) / float(size))
    out['g_rgb'] = t(rs.standard_normal((1, 3, res, res)) / float(res))
    return out


def stage2_fuse_state(template):
    """Fuse_sft_MLP parameters: N(0,1) / sqrt(fan-in) weights, N(0,1)-family biases (activations stay O(1))."""
    sd = {}
    for k, v in template.items():
        t = synthetic_tensor(STAGE2_FUSE_PREFIX + k, v.shape)
        sd[k] = t / np.sqrt(v.shape[1]) if k.endswith('weight') else t
    return sd


def stage2_head_state(template):
    """Texture-head parameters: a small FiLM perturbation (the reference zero-initialises fc_1, which would test nothing)."""
    return {k: 0.05 * synthetic_tensor(STAGE2_HEAD_PREFIX + k, v.shape) for k, v in template.items()}
