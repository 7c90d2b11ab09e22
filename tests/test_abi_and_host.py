"""CPU: the C-ABI library builds, loads and exports every symbol include/e3dge_hip.h declares; the ctypes
mirror of E3dgeRenderArgs matches the C layout; host-side logic (state-dict keys, adjoint geometry, camera
mirror, loud failure without a GPU)."""
import ctypes
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_golden

import e3dge_amd  # noqa: F401
from e3dge_amd import _lib, synthetic as syn

HEADER = os.path.join(REPO, "include", "e3dge_hip.h")


def test_library_exports_every_declared_symbol(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(e3dge_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in e3dge_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.e3dge_abi_version() == _lib.ABI_VERSION
    # fp32 fragment image + small blocks, the f16x3 (hi, lo) image of the same 64 chunks, the transposed fp32 and f16x3 images
    # ... and the 16x16x32 image of the 8-wave forward kernel
    assert lib.e3dge_siren_packed_floats() == (64 * 8192 + 2 * 1024 + 9 * 256 + 4 * 256 + 4) + 5 * 64 * 8192


def test_render_args_struct_layout_matches_c():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeRenderArgs), offsetof(E3dgeRenderArgs, sigmoid_beta),
         offsetof(E3dgeRenderArgs, batch), offsetof(E3dgeRenderArgs, force_background),
         offsetof(E3dgeRenderArgs, rgb), offsetof(E3dgeRenderArgs, dists));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    R = _lib.RenderArgs
    assert got == [ctypes.sizeof(R), R.sigmoid_beta.offset, R.batch.offset, R.force_background.offset,
                   R.rgb.offset, R.dists.offset]


def test_render_bwd_args_struct_layout_matches_c():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeRenderBwdArgs), offsetof(E3dgeRenderBwdArgs, d_rgb_map),
         offsetof(E3dgeRenderBwdArgs, tang), offsetof(E3dgeRenderBwdArgs, sigmoid_beta),
         offsetof(E3dgeRenderBwdArgs, force_background), offsetof(E3dgeRenderBwdArgs, dstyles),
         offsetof(E3dgeRenderBwdArgs, tex_alpha), offsetof(E3dgeRenderBwdArgs, d_tex_beta));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeSirenBwdArgs), offsetof(E3dgeSirenBwdArgs, tex_alpha),
         offsetof(E3dgeSirenBwdArgs, precision), offsetof(E3dgeSirenBwdArgs, n_pts), offsetof(E3dgeSirenBwdArgs, box_scale),
         offsetof(E3dgeSirenBwdArgs, d_tex_beta));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    R, Sb = _lib.RenderBwdArgs, _lib.SirenBwdArgs
    assert got == [ctypes.sizeof(R), R.d_rgb_map.offset, R.tang.offset, R.sigmoid_beta.offset,
                   R.force_background.offset, R.dstyles.offset, R.tex_alpha.offset, R.d_tex_beta.offset,
                   ctypes.sizeof(Sb), Sb.tex_alpha.offset, Sb.precision.offset, Sb.n_pts.offset, Sb.box_scale.offset,
                   Sb.d_tex_beta.offset]
    # argument validation of the training entry points happens before any launch
    lib = _lib.load()
    assert lib.e3dge_siren_render_bwd(None, None) == -1
    bad = _lib.RenderBwdArgs(batch=1, height=4, width=4, n_samples=0)
    assert lib.e3dge_siren_render_bwd(ctypes.byref(bad), None) == -1 and b"n_samples" in lib.e3dge_last_error()
    assert lib.e3dge_siren_bwd(None, None) == -1
    assert lib.e3dge_siren_bwd(ctypes.byref(_lib.SirenBwdArgs(batch=1, n_pts=10)), None) == -1 and b"null" in lib.e3dge_last_error()
    assert lib.e3dge_siren_sdf_grad(None, None, None, None, 1.0, 1, 10, None, None, 0, None) == -1
    assert lib.e3dge_siren_tangent(None, None, None, None, 1.0, 1, 10, None, 0, None) == -1
    assert lib.e3dge_siren_bwd_partial_floats(2, 1000) == 2 * 8 * 9 * 2 * 256     # 8 sub-tiles -> 8 workgroups per image


def test_modconv_structs_layout_matches_c():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(E3dgeModconvArgs), offsetof(E3dgeModconvArgs, out_amax), offsetof(E3dgeModconvArgs, negative_slope),
         offsetof(E3dgeModconvArgs, act), offsetof(E3dgeModconvArgs, noise_batch));
  printf("%zu %zu %zu %zu %zu %d\n", sizeof(E3dgeModLayer), offsetof(E3dgeModLayer, s_amax_out), offsetof(E3dgeModLayer, ci),
         offsetof(E3dgeModLayer, co_start), offsetof(E3dgeModLayer, lr_mul), E3DGE_AMAX_FLOATS);
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    A, L = _lib.ModconvArgs, _lib.ModLayer
    assert got == [ctypes.sizeof(A), A.out_amax.offset, A.negative_slope.offset, A.act.offset, A.noise_batch.offset,
                   ctypes.sizeof(L), L.s_amax_out.offset, L.ci.offset, L.co_start.offset, L.lr_mul.offset, _lib.AMAX_FLOATS]
    lib = _lib.load()
    assert lib.e3dge_modconv3x3(None, None) == -1
    bad = _lib.ModconvArgs(batch=1, ci=24, co=32, height=8, width=8)
    assert lib.e3dge_modconv3x3(ctypes.byref(bad), None) == -1
    assert lib.e3dge_modconv_packed_words(64, 32) == 2 * 2 * 9 * 2 * 64 * 4
    assert lib.e3dge_torgb(None, None, None, None, None, None, None, 1.0, 1, 32, 8, 6, None) == -1
    assert lib.e3dge_decoder_styles(None, 1, 1, 0, None, 1, 1, 1, None) == -1


def test_dec2_structs_layout_matches_c():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(E3dgeDec2Conv), offsetof(E3dgeDec2Conv, bias_amax), offsetof(E3dgeDec2Conv, noise_batch), sizeof(E3dgeDec2Rgb));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %d %zu %zu\n", sizeof(E3dgeDec2Plan), offsetof(E3dgeDec2Plan, conv1), offsetof(E3dgeDec2Plan, up),
         offsetof(E3dgeDec2Plan, rgb), offsetof(E3dgeDec2Plan, act), offsetof(E3dgeDec2Plan, amax), offsetof(E3dgeDec2Plan, negative_slope),
         offsetof(E3dgeDec2Plan, kernel_ms), E3DGE_DEC2_MAX_UP, offsetof(E3dgeDec2Plan, fir_blur_1d), offsetof(E3dgeDec2Plan, fir_blur_separable));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    C, R, P = _lib.Dec2Conv, _lib.Dec2Rgb, _lib.Dec2Plan
    assert got == [ctypes.sizeof(C), C.bias_amax.offset, C.noise_batch.offset, ctypes.sizeof(R), ctypes.sizeof(P), P.conv1.offset, P.up.offset,
                   P.rgb.offset, P.act.offset, P.amax.offset, P.negative_slope.offset, P.kernel_ms.offset, _lib.DEC2_MAX_UP,
                   P.fir_blur_1d.offset, P.fir_blur_separable.offset]
    lib = _lib.load()
    assert lib.e3dge_dec2_forward(None, None) == -1
    assert lib.e3dge_dec2_forward(ctypes.byref(P(batch=1, n_up=7, in_res=64, in_ch=256)), None) == -1       # more levels than the plan holds
    assert lib.e3dge_dec2_num_launches(4) == 22
    assert lib.e3dge_dec2_act_words(1, 32, 1024) == 4 * 2 * 1026 * 1026 * 4 and lib.e3dge_dec2_tbuf_floats(2, 32, 512) == 2 * 32 * 1027 * 1028
    assert lib.e3dge_dec2_pack(None, None, None, None, 1, 12, 8, None) == -1
    assert lib.e3dge_hitprob_points(None, None, None, None, None, None, None, None, 1, 4, 4, 1, None) == -1


def test_dec2_backward_structs_layout_matches_c():
    """ABI 12: the backward plan of the packed decoder pipeline (e3dge_dec2_backward) and the forward plan's new flag."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeDec2BwdConv), sizeof(E3dgeDec2BwdPlan), offsetof(E3dgeDec2BwdPlan, conv1),
         offsetof(E3dgeDec2BwdPlan, up), offsetof(E3dgeDec2BwdPlan, conv), offsetof(E3dgeDec2BwdPlan, gact), offsetof(E3dgeDec2BwdPlan, pbuf),
         offsetof(E3dgeDec2BwdPlan, drgb), offsetof(E3dgeDec2BwdPlan, bounds), offsetof(E3dgeDec2BwdPlan, kernel_ms),
         offsetof(E3dgeDec2BwdPlan, n_kernel_ms), offsetof(E3dgeDec2Plan, save_for_backward));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    C, Q, P = _lib.Dec2BwdConv, _lib.Dec2BwdPlan, _lib.Dec2Plan
    assert got == [ctypes.sizeof(C), ctypes.sizeof(Q), Q.conv1.offset, Q.up.offset, Q.conv.offset, Q.gact.offset, Q.pbuf.offset, Q.drgb.offset,
                   Q.bounds.offset, Q.kernel_ms.offset, Q.n_kernel_ms.offset, P.save_for_backward.offset]
    lib = _lib.load()
    assert lib.e3dge_dec2_backward(None, None, None) == -1
    plan = P(batch=1, n_up=1, in_res=8, in_ch=32)
    assert lib.e3dge_dec2_backward(ctypes.byref(plan), ctypes.byref(Q()), None) == -1        # the forward did not keep its top activation
    assert lib.e3dge_dec2_bwd_num_launches(4) == 21
    assert lib.e3dge_dec2_pbuf_words(1, 32, 1024) == 4 * 2 * 4 * 514 * 514 * 4 and lib.e3dge_dec2_pbuf_words(1, 32, 7) == 0
    assert lib.e3dge_dec2_prepack_weights_t(None, None, 1.0, 32, 32, 1, None) == -1


def test_ws_linear_struct_layout_matches_c_and_arguments_are_checked():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeWsLinear), offsetof(E3dgeWsLinear, n_rows), offsetof(E3dgeWsLinear, ld_x),
         offsetof(E3dgeWsLinear, off_y), offsetof(E3dgeWsLinear, post), offsetof(E3dgeWsLinear, w_fuse));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    W = _lib.WsLinear
    assert got == [ctypes.sizeof(W), W.n_rows.offset, W.ld_x.offset, W.off_y.offset, W.post.offset, W.w_fuse.offset]
    lib = _lib.load()
    assert lib.e3dge_ws_image_bytes(1) == 8 * 8 * 2 * 2 * 64 * 16 and lib.e3dge_ws_image_bytes(8) == 8 * 262144
    assert lib.e3dge_ws_linear(None, None) == -1
    one = ctypes.c_void_p(16)                                   # (never dereferenced: validation fails first)
    assert lib.e3dge_ws_linear(ctypes.byref(W(wimg=one, x=one, y=one, n_rows=4, ld_x=255, ld_y=256)), None) == -1      # 256 columns do not fit
    assert lib.e3dge_ws_linear(ctypes.byref(W(wimg=one, x=one, y=one, n_rows=4, ld_x=256, ld_y=256, post=2)), None) == -1  # fuse without D, S
    assert lib.e3dge_ws_linear(ctypes.byref(W(wimg=one, x=one, y=one, n_rows=4, ld_x=256, ld_y=256, colw=one)), None) == -1  # colw without m
    assert lib.e3dge_ws_linear(ctypes.byref(W(wimg=one, x=one, y=one, n_rows=0, ld_x=256, ld_y=256)), None) == 0          # nothing to do
    assert lib.e3dge_ws_pack(None, None, 1, None) == -1


def test_wgrad_and_texhead_bwd_layout_and_argument_checks():
    """Round 5: struct E3dgeWgrad against the ctypes mirror (compiled C probe), the workspace formulas, and validation before any launch."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "e3dge_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(E3dgeWgrad), offsetof(E3dgeWgrad, c), offsetof(E3dgeWgrad, ws_floats), offsetof(E3dgeWgrad, n_rows),
         offsetof(E3dgeWgrad, lda), offsetof(E3dgeWgrad, ldc), offsetof(E3dgeWgrad, relu_b), offsetof(E3dgeWgrad, xcol), offsetof(E3dgeWgrad, ccol),
         offsetof(E3dgeWgrad, ld_xcol), offsetof(E3dgeWgrad, b_gap));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    G = _lib.Wgrad
    assert got == [ctypes.sizeof(G), G.c.offset, G.ws_floats.offset, G.n_rows.offset, G.lda.offset, G.ldc.offset, G.relu_b.offset, G.xcol.offset, G.ccol.offset,
                   G.ld_xcol.offset, G.b_gap.offset]
    lib = _lib.load()
    # round 6: blocks of 256 x 256.  98,304 points, 512 x 301 outputs: 2 x 2 blocks, 256 // 4 = 64 slabs; behind the partial blocks the column
    # partials (2 x 256 per slab and block row).  256 x 256: one block, 256 slabs.
    assert lib.e3dge_wgrad_ws_floats(512, 301, 98304) == 64 * 2 * (2 * 256 * 256 + 2 * 256)
    assert lib.e3dge_wgrad_ws_floats(256, 256, 98304) == 256 * (256 * 256 + 2 * 256)
    assert lib.e3dge_wgrad_ws_floats(5, 3, 1) == 256 * 256 + 2 * 256 and lib.e3dge_wgrad_ws_floats(0, 3, 10) == 0
    assert lib.e3dge_wgrad(None, None) == -1
    one = ctypes.c_void_p(16)
    assert lib.e3dge_wgrad(ctypes.byref(G(a=one, amax_a=one, b=one, amax_b=one, c=one, ws=one, n_rows=4, lda=3, m=4, ldb=4, n=4, ldc=4, ws_floats=1 << 20)), None) == -1   # m > lda
    assert lib.e3dge_wgrad(ctypes.byref(G(a=one, amax_a=one, b=one, amax_b=one, c=one, ws=one, n_rows=4, lda=4, m=4, ldb=4, n=4, ldc=4, ws_floats=16)), None) == -1        # workspace
    assert lib.e3dge_wgrad(ctypes.byref(G(c=one, n_rows=4, m=0, n=4, ldc=4)), None) == 0                                                                                # nothing to do
    ok = dict(a=one, amax_a=one, b=one, amax_b=one, c=one, ws=one, n_rows=4, lda=4, m=4, ldb=600, n=512, ldc=600, ws_floats=1 << 30)
    assert lib.e3dge_wgrad(ctypes.byref(G(b_gap=1, b_gap_at=100, **ok)), None) == -1                    # the gap starts at a multiple of 256
    assert lib.e3dge_wgrad(ctypes.byref(G(xcol=one, ld_xcol=1, **ok)), None) == -1                      # xcol without ccol
    assert lib.e3dge_tex_modulations_bwd_ws_floats(1000) == 2 * 1000 * 320 and lib.e3dge_tex_modulations_bwd_ws_floats(0) == 0
    assert lib.e3dge_resblock_bwd_packed_floats() == 120 * 5120 + 320
    assert lib.e3dge_tex_modulations_bwd(None, None, 301, 5, None, None, None, None, None, None, None) == -1
    assert lib.e3dge_tex_modulations_bwd(one, one, 321, 5, one, one, one, one, None, None, None) == -1       # cin > 320
    assert lib.e3dge_tex_modulations_bwd(None, None, 301, 0, None, None, None, None, None, None, None) == 0   # nothing to do
    assert lib.e3dge_resblock_bwd_pack_weights(None, None, None, None, None, 301, None) == -1


def test_wgrad_host_side_rules(monkeypatch):
    """e3dge_amd.wgrad: the library backend is plain matmul (CPU tensors included); the native one refuses CPU tensors and odd layouts loudly."""
    import torch
    from e3dge_amd import wgrad as W
    a, b = torch.randn(7, 3), torch.randn(7, 5)
    monkeypatch.setenv("E3DGE_WGRAD", "library")
    assert torch.allclose(W.wgrad(a, b, relu_b=True), a.t() @ torch.relu(b))
    out = torch.zeros(3, 9)
    W.wgrad(a, b, out=out[:, 2:7])
    assert torch.allclose(out[:, 2:7], a.t() @ b) and float(out[:, :2].abs().max()) == 0.0
    monkeypatch.setenv("E3DGE_WGRAD", "hip")
    with pytest.raises(RuntimeError, match="GPU"):
        W.wgrad(a, b)
    with pytest.raises(ValueError, match="unit column stride"):
        W.wgrad(a.t().contiguous().t(), b)
    monkeypatch.setenv("E3DGE_WGRAD", "fast")
    with pytest.raises(ValueError, match="E3DGE_WGRAD"):
        W.wgrad(a, b)


def test_points_launch_geometry_minimises_the_span_in_256_cu_rounds(lib):
    """Round 5: the sub-tiles a workgroup of a points launch walks are chosen by the launch's span (rounds of 256 workgroups x sub-tiles), the
    largest count among the shortest spans.  Visible through the backward's partial-sum workspace = batch x workgroups per image x 9 x 2 x 256."""
    per_wg = 9 * 2 * 256                                            # (round 6: one slice per 128-point sub-tile -- workgroups x sub-tiles per workgroup)
    tiles = 73728 // 128                                            # 64 x 64 x 18 samples: 576 tiles of 128 points
    assert lib.e3dge_siren_bwd_partial_floats(1, 73728) == 1 * (tiles // 3) * 3 * per_wg      # 192 workgroups x 3 (one round; 64 CUs stay free)
    assert lib.e3dge_siren_bwd_partial_floats(2, 73728) == 2 * 116 * 5 * per_wg               # 5 sub-tiles: 232 workgroups, one round of 5
    # four samples: 8 sub-tiles would be 288 workgroups = TWO rounds of 8 for 9 rounds of work (the rule until round 5); 3 x 768 = three full rounds
    assert lib.e3dge_siren_bwd_partial_floats(4, 73728) == 4 * 192 * 3 * per_wg
    assert lib.e3dge_siren_bwd_partial_floats(8, 73728) == 8 * 96 * 6 * per_wg                # 6 sub-tiles: 768 workgroups, three rounds of 6 = 18
    assert lib.e3dge_siren_bwd_partial_floats(1, 4096) == 32 * per_wg                         # the surface points: one tile per workgroup
    assert lib.e3dge_siren_bwd_partial_floats(1, 98304) == 256 * 3 * per_wg                       # 768 tiles: 256 workgroups x 3


def test_round5_backend_switches_are_validated(monkeypatch):
    """The A/B switches of round 5 accept their documented values only (a typo must not silently select a path)."""
    from e3dge_amd import stylesdf_model as sm, volume_renderer as vr
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "library")
    assert vr.tex_head_backward_backend() == "library"
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "Hip")
    assert vr.tex_head_backward_backend() == "hip"
    monkeypatch.setenv("E3DGE_TEXHEAD_BWD", "fast")
    with pytest.raises(ValueError, match="E3DGE_TEXHEAD_BWD"):
        vr.tex_head_backward_backend()
    monkeypatch.delenv("E3DGE_DEC2_DLATENT", raising=False)
    assert sm.decoder_dlatent_native() is True
    monkeypatch.setenv("E3DGE_DEC2_DLATENT", "0")
    assert sm.decoder_dlatent_native() is False
    monkeypatch.setenv("E3DGE_DECODER_AUTOGRAD", "fast")
    with pytest.raises(RuntimeError, match="E3DGE_DECODER_AUTOGRAD"):
        sm.decoder_autograd_backend()


def test_host_helpers_that_need_no_gpu(lib):
    assert lib.e3dge_upfirdn2d_out_size(129, 1, 1, 1, 1, 4) == 128     # Blur after the 64->129 transposed conv
    assert lib.e3dge_upfirdn2d_out_size(64, 2, 1, 2, 1, 4) == 128      # skip Upsample
    assert lib.e3dge_upfirdn2d_out_size(3, 1, 1, 0, 0, 4) < 0          # empty output is an error, not a crash
    # argument validation happens before any launch -> testable without a GPU
    assert lib.e3dge_upfirdn2d(None, None, None, 1, 8, 8, 40, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == -1
    assert b"kernel" in lib.e3dge_last_error()
    assert lib.e3dge_fused_bias_act(None, None, None, None, 3, 0, 0.2, 1.0, 1 << 31, 1, 0, None) == -1
    args = _lib.RenderArgs(n_samples=8, batch=1, height=4, width=4)
    assert lib.e3dge_siren_render_fwd(ctypes.byref(args), None) == -1


def test_state_dict_keys_equal_the_reference_generator():
    from e3dge_amd.stylesdf_model import G_pred_latents
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys_1024.json")))
    g = G_pred_latents(syn.model_opt(), syn.rendering_opt(), full_pipeline=True)
    mine = {k: list(v.shape) for k, v in g.state_dict().items()}
    assert mine == ref, (sorted(set(mine) ^ set(ref))[:10])


def test_local_global_wrapper_keeps_netglobal_keys():
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    r = VolumeFeatureRenderer(syn.rendering_opt(enable_local_model=True), mode='test')
    keys = list(r.state_dict())
    assert 'sigmoid_beta' in keys and 'network.netGlobal.pts_linears.0.gamma.weight' in keys
    assert 'network.netGlobal.views_linears.weight' in keys and 'network.netGlobal.sigma_linear.bias' in keys
    assert not any('netLocal' in k for k in keys)
    # with the texture head enabled the reference's key path of the ResnetBlockFC appears, zero-initialised like there
    r2 = VolumeFeatureRenderer(syn.rendering_opt(enable_local_model=True, L_pred_tex_modulations=True), mode='test')
    pre = 'network.netLocal.local_feat_to_tex_modulations_linear.'
    shapes = {k[len(pre):]: tuple(v.shape) for k, v in r2.state_dict().items() if k.startswith(pre)}
    assert shapes == {'fc_0.weight': (301, 301), 'fc_0.bias': (301,), 'fc_1.weight': (512, 301), 'fc_1.bias': (512,),
                      'shortcut.weight': (512, 301)}
    assert all(float(v.abs().max()) == 0 for k, v in r2.state_dict().items() if k.startswith(pre))
    assert lib_floats() == 84 * 5120 + 320 + 512 + 4          # 84 weight chunks of 20 KiB, b_0, b_1, 4 bound scalars


def lib_floats():
    return _lib.load().e3dge_resblock_packed_floats()


def test_no_cpu_fallback_on_the_product_path():
    from e3dge_amd import op
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    x = torch.randn(1, 2, 8, 8)
    with pytest.raises(RuntimeError, match="GPU"):           # the raw kernel entry points have no CPU path
        op.fused_bias_act(x, torch.zeros(2), None, 3, 0, 0.2, 1.0)
    with pytest.raises(RuntimeError, match="GPU"):
        op.upfirdn2d_raw(x.reshape(2, 8, 8, 1), torch.ones(4, 4), 1, 1, 1, 1, 0, 0, 0, 0)
    r = VolumeFeatureRenderer(syn.rendering_opt(), mode='test')
    cam = torch.zeros(1, 3, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        r(cam, torch.ones(1, 1, 1), torch.ones(1, 1, 1), torch.ones(1, 1, 1), styles=torch.zeros(1, 9, 256))


def test_op_cpu_branches_match_the_reference_vectors():
    """`op.fused_leaky_relu` / `op.upfirdn2d` accept CPU tensors like the reference's wrappers do (fused_act.py:107-118,
    upfirdn2d.py:146-200): own plain-torch code, checked against the vectors recorded from the reference, first-order
    autograd included."""
    from e3dge_amd import op
    g = load_golden("upfirdn2d")
    for name in ('blur_up', 'upsample', 'downsample', 'blur_down', 'k3', 'crop', 'big'):
        x = torch.from_numpy(g[name + '_x']).requires_grad_(True)
        cfg = [int(v) for v in g[name + '_cfg']]
        y = op.upfirdn2d(x, torch.from_numpy(g[name + '_k']), up=cfg[0], down=cfg[1], pad=(cfg[2], cfg[3]))
        gx, = torch.autograd.grad(y, x, torch.from_numpy(g[name + '_gy']))
        np.testing.assert_allclose(y.detach().numpy(), g[name + '_y'], atol=2e-6)
        np.testing.assert_allclose(gx.numpy(), g[name + '_gx'], atol=2e-6)
    a = load_golden("fused_act")
    for name in ('conv', 'mapping', 'nobias', 'ragged'):
        x = torch.from_numpy(a[name + '_x']).requires_grad_(True)
        b = torch.from_numpy(a[name + '_b']).requires_grad_(True) if name + '_b' in a.files else None
        y = op.fused_leaky_relu(x, b, 0.2, float(a[name + '_scale']))
        grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), torch.from_numpy(a[name + '_gy']))
        np.testing.assert_allclose(y.detach().numpy(), a[name + '_y'], atol=2e-6)
        np.testing.assert_allclose(grads[0].numpy(), a[name + '_gx'], atol=2e-6)
        if b is not None:
            np.testing.assert_allclose(grads[1].numpy(), a[name + '_gb'], rtol=1e-5, atol=1e-5)
    m = op.FusedLeakyReLU(3)
    assert tuple(m(torch.randn(2, 3, 4, 4)).shape) == (2, 3, 4, 4)
    with pytest.raises(RuntimeError):
        op.upfirdn2d(torch.zeros(1, 1, 2, 2), torch.ones(4, 4))       # empty output is an error here too


def test_unsupported_options_fail_loudly():
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    with pytest.raises(NotImplementedError):
        VolumeFeatureRenderer(syn.rendering_opt(no_sdf=True))
    with pytest.raises(NotImplementedError):
        VolumeFeatureRenderer(syn.rendering_opt(depth=6))
    r = VolumeFeatureRenderer(syn.rendering_opt(), mode='test')
    with pytest.raises(RuntimeError, match="no CPU path"):        # the product path never falls back to the CPU
        r(None, None, None, None, styles=None, return_mesh=True)


def test_upfirdn_adjoint_geometry_is_an_involution():
    from e3dge_amd.op.upfirdn2d import Geometry
    for (h, w, k, up, down, pad) in [(33, 33, 4, 1, 1, (1, 1)), (16, 16, 4, 2, 1, (2, 1)), (32, 32, 4, 1, 2, (1, 1)),
                                     (19, 23, 4, 2, 1, (-1, 2))]:
        g = Geometry((up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
        oh, ow = g.out_hw(h, w, k, k)
        a = g.adjoint(h, w, k, k)
        assert a.out_hw(oh, ow, k, k) == (h, w)                        # the adjoint maps back to the input extent
        aa = a.adjoint(oh, ow, k, k)
        assert aa.up == g.up and aa.down == g.down and aa.pad[0] == g.pad[0] and aa.pad[2] == g.pad[2]


def test_camera_mirror_matches_reference_vectors():
    from e3dge_amd.camera_utils import generate_camera_params
    g = load_golden("camera")
    cam = generate_camera_params(64, 'cpu', locations=torch.from_numpy(g['locations']), return_calibs=True)
    np.testing.assert_allclose(cam['poses'].numpy(), g['ref_poses'], atol=1e-6)
    np.testing.assert_allclose(cam['focal'].numpy(), g['ref_focal'], rtol=1e-6)
    np.testing.assert_allclose(cam['calibs'].numpy(), g['ref_calibs'], atol=1e-5)
    poses, focal, near, far, _ = generate_camera_params(128, 'cpu', locations=torch.from_numpy(g['traj_locations']))
    np.testing.assert_allclose(poses.numpy(), g['ref_traj_poses'], atol=1e-6)
    np.testing.assert_allclose(near.numpy().ravel(), 0.88, atol=1e-6)


def test_synthetic_weights_are_key_deterministic():
    a = syn.synthetic_tensor('renderer.network.pts_linears.3.weight', (256, 256))
    b = syn.synthetic_tensor('renderer.network.pts_linears.3.weight', (256, 256))
    c = syn.synthetic_tensor('renderer.network.pts_linears.4.weight', (256, 256))
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.abs().max()) <= np.sqrt(6 / 256) / 25 + 1e-9


def test_modules_are_deep_copyable():
    """Runners copy generators (EMA copies, `surface_g_ema`): no stream / ctypes / lambda state may sit on the modules."""
    import copy
    from e3dge_amd.stylesdf_model import G_pred_latents
    g = G_pred_latents(syn.model_opt(size=128, channel_multiplier=1, renderer_spatial_output_dim=16), syn.rendering_opt(),
                       full_pipeline=True)
    g2 = copy.deepcopy(g)
    assert g2.state_dict().keys() == g.state_dict().keys()
    assert g2.renderer.opt.N_samples == g.renderer.opt.N_samples and g2.renderer.opt.no_such_option is None


def test_params_of_notices_added_removed_and_replaced_parameters():
    """_lib.params_of caches the (sub-module, name) slots of a module; the weight-image caches and the requires_grad checks of the launch
    wrappers are keyed on what it returns, so it must keep matching module.parameters() through module surgery (round-4 advisor finding)."""
    import copy
    import torch
    from torch import nn
    m = nn.Sequential(nn.Linear(3, 4), nn.Linear(4, 2))
    same = lambda mod: [id(p) for p in _lib.params_of(mod)] == [id(p) for p in mod.parameters()]
    assert same(m) and same(m)
    m[0].weight = nn.Parameter(torch.zeros(4, 3))                  # replaced object, same slot
    assert same(m)
    m.add_module("extra", nn.Linear(2, 2))                         # sub-module added later
    assert same(m) and len(_lib.params_of(m)) == 6
    m[1].register_parameter("gain", nn.Parameter(torch.ones(1)))   # parameter added later
    assert same(m) and len(_lib.params_of(m)) == 7
    del m[1]._parameters["gain"]                                   # parameter removed (weight_norm / parametrize do this): no KeyError
    assert same(m) and len(_lib.params_of(m)) == 6
    m[0].bias = None
    assert same(m) and len(_lib.params_of(m)) == 5
    twin = copy.copy(m)                                            # a shallow copy (DataParallel's replicas) with its own parameter dicts
    twin._modules = {k: copy.copy(v) for k, v in m._modules.items()}
    for v in twin._modules.values():
        v._parameters = {k: nn.Parameter(p.detach().clone()) for k, p in v._parameters.items() if p is not None}
    assert same(twin) and same(m)
    assert same(copy.deepcopy(m))


def test_decoder_autograd_dispatch_rules_on_the_host(monkeypatch):
    """Which forwards may take the packed decoder inside an autograd graph (round 5, host logic only -- no GPU needed to decide):
    the packed backward exists for d features only, needs 32-channel multiples on both sides of every 3x3 layer and 64 input channels
    on the up-sampling ones; the mode switch rejects unknown values; CPU tensors never take the packed path."""
    import torch
    from e3dge_amd import synthetic as syn
    from e3dge_amd import stylesdf_model as sm
    for v in ("auto", "packed", "library"):
        monkeypatch.setenv("E3DGE_DECODER_AUTOGRAD", v)
        assert sm.decoder_autograd_backend() == v
    monkeypatch.setenv("E3DGE_DECODER_AUTOGRAD", "fast")
    with pytest.raises(RuntimeError):
        sm.decoder_autograd_backend()
    monkeypatch.delenv("E3DGE_DECODER_AUTOGRAD")
    assert sm.decoder_autograd_backend() == "auto"

    def dec(size, cm, res=64):
        g = sm.G_pred_latents(syn.model_opt(size=size, channel_multiplier=cm, renderer_spatial_output_dim=res), syn.rendering_opt(), full_pipeline=True)
        return g.decoder
    assert dec(1024, 2)._dec2_bwd_ok()                       # 512 / 256 / 128 / 64 / 32 channels
    assert dec(256, 1)._dec2_bwd_ok()                        # 256 / 128 / 64
    assert not dec(1024, 1)._dec2_bwd_ok()                   # the last level has 16 channels: not a 32-channel multiple
    monkeypatch.setenv("E3DGE_DEC2_BWD", "library")
    assert not dec(256, 1)._dec2_bwd_ok()
    monkeypatch.delenv("E3DGE_DEC2_BWD")
    d = dec(256, 1)
    f = torch.zeros(1, 256, 64, 64, requires_grad=True)
    lat = torch.zeros(1, d.n_latent, 512)
    assert d._needs_graph(f, lat) and not d._dec2_ok(f, lat, [None] * d.num_layers, None)      # CPU tensors: never the packed path
    with torch.no_grad():
        assert not d._needs_graph(f, lat)
