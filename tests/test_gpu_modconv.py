"""GPU: the fused modulated 3x3 convolution (e3dge_modconv3x3) against the oracle's restatement of
ModulatedConv2d / StyledConv (oracle/decoder_ref.py, bit-identical to the reference on the decoder goldens), and against
the float64 evaluation of the same.

Stated fp32 tolerance: outputs are O(1) (demodulated weights have unit norm); the fp32 oracle itself sits ~1e-6 from
float64.  Bound: |hip - oracle| <= 2e-5 * max(1, max|out|) and |hip - f64| <= 3x the fp32 oracle's own distance + 1e-6*max|out|
-- the split-f16 contraction (hi+lo operands, three products, fp32 accumulate) is as accurate as an fp32 convolution."""
import math

import numpy as np
import pytest
import torch

from conftest import maxerr, record
from oracle import decoder_ref

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn
from e3dge_amd.stylesdf_model import ModulatedConv2d, StyledConv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_layer(cls, ci, co, upsample, seed, prefix='decoder.convs.0.'):
    m = cls(ci, co, 3, 512, upsample=upsample)
    sd = {k: syn.synthetic_tensor(prefix + k, v.shape, seed) for k, v in m.state_dict().items() if not k.endswith('kernel')}
    m.load_state_dict(sd, strict=False)
    return m.to(DEV).eval(), {prefix + k: v for k, v in sd.items()}


CASES = [  # ci, co, H, W, B, upsample
    (16, 32, 8, 8, 1, False), (32, 64, 20, 36, 2, False), (64, 64, 64, 64, 1, False), (48, 96, 33, 70, 1, False),
    (256, 128, 16, 16, 2, False), (64, 32, 128, 128, 1, False), (128, 128, 130, 66, 1, False),
    (16, 32, 8, 8, 1, True), (32, 64, 20, 36, 2, True), (128, 64, 64, 64, 1, True), (64, 32, 70, 130, 1, True),
    (48, 96, 9, 150, 1, True),
]


@pytest.mark.parametrize("ci,co,H,W,B,upsample", CASES)
def test_modulated_conv_vs_oracle(ci, co, H, W, B, upsample):
    m, sd = make_layer(ModulatedConv2d, ci, co, upsample, seed=ci + co, prefix='decoder.convs.0.conv.')
    rs = np.random.RandomState(H * W + ci)
    x = torch.from_numpy((rs.standard_normal((B, ci, H, W)) * (0.2 + 2 * rs.uniform(size=(1, ci, 1, 1)))).astype(np.float32)).to(DEV)
    style = torch.from_numpy(rs.standard_normal((B, 512)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        assert m.fused_ok(x)
        y = m(x, style)
        ref = decoder_ref.modulated_conv(sd, 'decoder.convs.0.conv.', x.cpu(), style.cpu(), True, upsample)
        t64 = decoder_ref.modulated_conv({k: v.double() for k, v in sd.items()}, 'decoder.convs.0.conv.', x.cpu().double(),
                                         style.cpu().double(), True, upsample)
    scale = max(1.0, float(t64.abs().max()))
    e = dict(vs_oracle=maxerr(y, ref), vs_f64=maxerr(y, t64), oracle_vs_f64=maxerr(ref, t64), out_max=scale)
    record(f"modconv_{ci}_{co}_{H}x{W}_b{B}_up{int(upsample)}", **e)
    assert tuple(y.shape) == tuple(ref.shape)
    assert e['vs_oracle'] <= 2e-5 * scale and e['vs_f64'] <= 3 * e['oracle_vs_f64'] + 1e-6 * scale, e


@pytest.mark.parametrize("ci,co,H,W,B", [(32, 32, 24, 40, 2), (64, 128, 64, 64, 1)])
def test_styled_conv_fused_tail(ci, co, H, W, B):
    """Stride-1 StyledConv: conv + noise + bias + lrelu in ONE launch; fixed noise per sample and shared noise."""
    m, sd = make_layer(StyledConv, ci, co, False, seed=7)
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, ci, H, W)).astype(np.float32)).to(DEV)
    style = torch.from_numpy(rs.standard_normal((B, 512)).astype(np.float32)).to(DEV)
    for nb in (1, B):
        noise = torch.from_numpy(rs.standard_normal((nb, 1, H, W)).astype(np.float32)).to(DEV)
        with torch.no_grad():
            y = m(x, style, noise=noise)
            ref = decoder_ref.styled_conv(sd, 'decoder.convs.0.', x.cpu(), style.cpu(), noise.cpu())
        assert maxerr(y, ref) <= 2e-5 * max(1.0, float(ref.abs().max())), (nb, maxerr(y, ref))
    with torch.no_grad():
        a, b = m(x, style), m(x, style)                     # random noise: runs, differs between calls
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_operand_scaling_is_magnitude_invariant():
    """The activations are scaled by a power of two derived from max|s|*max|x|: the relative error must not depend on the
    input magnitude (1e-6 ... 1e+5), and huge inputs must not overflow the f16 operands."""
    m, sd = make_layer(ModulatedConv2d, 32, 32, False, seed=2, prefix='decoder.convs.0.conv.')
    rs = np.random.RandomState(1)
    x0 = rs.standard_normal((1, 32, 32, 32)).astype(np.float32)
    style = torch.from_numpy(rs.standard_normal((1, 512)).astype(np.float32)).to(DEV)
    for mag in (1e-6, 1.0, 1e5):
        x = torch.from_numpy(x0 * np.float32(mag)).to(DEV)
        with torch.no_grad():
            y = m(x, style)
            t64 = decoder_ref.modulated_conv({k: v.double() for k, v in sd.items()}, 'decoder.convs.0.conv.', x.cpu().double(),
                                             style.cpu().double(), True, False)
        rel = maxerr(y, t64) / float(t64.abs().max())
        record(f"modconv_magnitude_{mag:g}", rel_err=rel)
        assert torch.isfinite(y).all() and rel <= 2e-6, (mag, rel)
    z = torch.zeros(1, 32, 32, 32, device=DEV)
    with torch.no_grad():
        assert float(m(z, style).abs().max()) == 0.0


def test_backends_agree_and_fused_is_not_slower():
    """The 1024^2 decoder's eight 3x3 layers on both backends (fused kernel vs e3dge_modconv_weights + library convolution):
    same results; timings recorded (informational: a slower fused path is a warning, not a failure)."""
    import os
    import time
    shapes = [(256, 512, 64, False), (512, 256, 64, True), (256, 256, 128, False), (256, 128, 128, True), (128, 128, 256, False),
              (128, 64, 256, True), (64, 64, 512, False), (64, 32, 512, True), (32, 32, 1024, False)]
    tot = dict(hip=0.0, library=0.0)
    rows = {}
    for ci, co, res, up in shapes:
        m, _ = make_layer(StyledConv, ci, co, up, seed=ci)
        x = torch.randn(1, ci, res, res, device=DEV)
        style = torch.randn(1, 512, device=DEV)
        ores = 2 * res if up else res
        noise = torch.randn(1, 1, ores, ores, device=DEV)
        out = {}
        for be in ("hip", "library"):
            os.environ["E3DGE_MODCONV"] = be
            try:
                with torch.no_grad():
                    for _ in range(2):
                        y = m(x, style, noise=noise)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        y = m(x, style, noise=noise)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / 5 * 1e3
            finally:
                os.environ.pop("E3DGE_MODCONV", None)
            out[be] = (y, ms)
            tot[be] += ms
        err = maxerr(out["hip"][0], out["library"][0])
        gflop = 2 * 9 * ci * co * res * res / 1e9
        rows[f"{ci}->{co}@{res}{'up' if up else ''}"] = dict(hip_ms=out["hip"][1], library_ms=out["library"][1], max_abs_diff=err,
                                                            hip_tflops=gflop / out["hip"][1])
        assert err <= 5e-5 * max(1.0, float(out["library"][0].abs().max())), (ci, co, res, up, err)
    record("modconv_decoder_layers", total_hip_ms=tot["hip"], total_library_ms=tot["library"],
           **{k + "_" + kk: vv for k, v in rows.items() for kk, vv in v.items()})
    # (wall-clock over five calls per layer: a number for the report, not a gate -- one disturbed call on a shared box must not fail the
    # parity suite; the decoder's timings are measured properly by bench.py and tools/dec2_check.py)
    if not tot["hip"] < 1.5 * tot["library"]:
        import warnings
        warnings.warn(f"fused modulated convolutions slower than the library path in this run: {tot}")


@pytest.mark.parametrize("ci,res,B,with_skip", [(32, 16, 2, True), (512, 64, 1, False), (64, 128, 1, True), (48, 36, 2, True)])
def test_torgb_fused(ci, res, B, with_skip):
    """ToRGB in one launch (1x1 modulated conv without demodulation + bias + FIR-up-sampled skip) against the oracle."""
    from e3dge_amd.stylesdf_model import ToRGB
    m = ToRGB(ci, 512, upsample=True)
    sd = {k: syn.synthetic_tensor('decoder.to_rgbs.0.' + k, v.shape, ci) for k, v in m.state_dict().items() if not k.endswith('kernel')}
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV).eval()
    sdp = {'decoder.to_rgbs.0.' + k: v for k, v in sd.items()}
    rs = np.random.RandomState(res)
    x = torch.from_numpy(rs.standard_normal((B, ci, res, res)).astype(np.float32)).to(DEV)
    style = torch.from_numpy(rs.standard_normal((B, 512)).astype(np.float32)).to(DEV)
    skip = torch.from_numpy(rs.standard_normal((B, 3, res // 2, res // 2)).astype(np.float32)).to(DEV) if with_skip else None
    with torch.no_grad():
        assert m.fused_ok(x, skip)
        y = m(x, style, skip=skip)
        ref = decoder_ref.to_rgb(sdp, 'decoder.to_rgbs.0.', x.cpu(), style.cpu(), None if skip is None else skip.cpu())
    e = maxerr(y, ref)
    record(f"torgb_{ci}_{res}_b{B}_skip{int(with_skip)}", max_abs_err=e, out_max=float(ref.abs().max()))
    assert e <= 1e-5 * max(1.0, float(ref.abs().max())), e


def test_up_layer_fused_tail_and_amax_tracking():
    """Up-sampling StyledConv: transposed conv by phase -> blur + noise + bias + lrelu in one pass; the amax buffer it leaves
    equals max|output| (the next layer's operand scale comes from it)."""
    from e3dge_amd import _lib
    m, sd = make_layer(StyledConv, 64, 32, True, seed=9)
    rs = np.random.RandomState(8)
    x = torch.from_numpy(rs.standard_normal((2, 64, 24, 40)).astype(np.float32)).to(DEV)
    style = torch.from_numpy(rs.standard_normal((2, 512)).astype(np.float32)).to(DEV)
    noise = torch.from_numpy(rs.standard_normal((1, 1, 48, 80)).astype(np.float32)).to(DEV)
    am = torch.zeros(_lib.AMAX_FLOATS, device=DEV)
    with torch.no_grad():
        y = m(x, style, noise=noise, out_amax=am)
        ref = decoder_ref.styled_conv(sd, 'decoder.convs.0.', x.cpu(), style.cpu(), noise.cpu(), upsample=True)
    assert maxerr(y, ref) <= 2e-5 * max(1.0, float(ref.abs().max())), maxerr(y, ref)
    assert float(am.max()) == float(y.abs().max())
    m2, sd2 = make_layer(StyledConv, 32, 32, False, seed=10)
    am2 = torch.zeros(_lib.AMAX_FLOATS, device=DEV)
    with torch.no_grad():
        z = m2(y, style, noise=noise, in_amax=am, out_amax=am2)
        ref2 = decoder_ref.styled_conv(sd2, 'decoder.convs.0.', y.cpu(), style.cpu(), noise.cpu())
    assert maxerr(z, ref2) <= 2e-5 * max(1.0, float(ref2.abs().max())) and float(am2.max()) == float(z.abs().max())
