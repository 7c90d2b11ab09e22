"""Parameter gradients of the local branch's nn.Linear layers through e3dge_wgrad (csrc/wgrad.hip): `grad_output.t() @ f(input)`, f = identity or
relu, contracted over the points in split-f16 x 3 MFMAs with a fixed-order split-K fold -- what autograd runs for ResnetBlockFC
(helper_modules/resnetfc.py:49-58) and Fuse_sft_MLP (helper_modules/sft.py:84-110) in the reference's stage-2 step (e3dge_full_runner.py:185-317).
E3DGE_WGRAD = hip (default) | library (torch matmul, rounds 2-4)."""
import ctypes
import os

import torch

from . import _lib


def wgrad_backend():
    v = os.environ.get("E3DGE_WGRAD", "hip").lower()
    if v not in ("hip", "library"):
        raise ValueError(f"E3DGE_WGRAD={v!r}: expected hip or library")
    return v


def amax_of(t):
    """An amax buffer (include/e3dge_hip.h) bounding |t|: e3dge_amax over the tensor when it is contiguous, over its rows' span otherwise."""
    am = torch.zeros(_lib.AMAX_FLOATS, device=t.device, dtype=torch.float32)
    if t.numel() == 0:
        return am
    if t.is_contiguous() and t.data_ptr() % 16 == 0:
        with _lib.on_device(t.device):
            _lib.check(_lib.load().e3dge_amax(_lib.ptr(am), _lib.ptr(t), t.numel(), _lib.stream_of(t)), "e3dge_amax")
    else:
        am[0] = t.abs().amax()
    return am


def _rows(t, what):
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"wgrad: {what} must be a 2-D fp32 tensor with unit column stride (got shape {tuple(t.shape)}, strides {t.stride()}, {t.dtype})")
    _lib.require_gpu(t, what)
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def wgrad(a, b, relu_b=False, amax_a=None, amax_b=None, out=None, colsum=False, gap_col=None):
    """a (P, m), b (P, n): fp32 GPU row tensors (any row pitch) -> a^T f(b) as an (m, n) tensor (or into `out`, a row tensor of pitch >= n).
    colsum=True: returns (a^T f(b), a.sum(0)) -- the layer's bias gradient from the same pass over the rows (round 6).
    gap_col (a multiple of 256, < n): column `gap_col` of b is not part of the block grid; it is contracted as a weighted column sum in
    the same launch (the visibility-mask column of Fuse_sft_MLP's 513-wide input) -- same result, n - 1 columns of MFMA work."""
    if wgrad_backend() == "library":
        r = a.t() @ (torch.relu(b) if relu_b else b)
        if out is not None:
            out.copy_(r)
            r = out
        return (r, a.sum(0)) if colsum else r
    lda, ldb = _rows(a, "a"), _rows(b, "b")
    P, m = a.shape
    n = b.shape[1]
    if b.shape[0] != P:
        raise ValueError(f"wgrad: a has {P} rows, b {b.shape[0]}")
    if gap_col is not None and not (0 < gap_col < n and gap_col % 256 == 0):
        raise ValueError(f"wgrad: gap_col={gap_col} must be a multiple of 256 inside b's {n} columns")
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    ldc = _rows(out, "out")
    cs = torch.empty(m, device=a.device, dtype=torch.float32) if colsum else None
    if m == 0 or n == 0:
        return (out, cs) if colsum else out
    lib = _lib.load()
    amax_a = amax_of(a) if amax_a is None else amax_a
    amax_b = amax_of(b) if amax_b is None else amax_b
    n_eff = n - 1 if gap_col is not None else n
    n_ws = lib.e3dge_wgrad_ws_floats(m, n_eff, P)
    ws = torch.empty(max(n_ws, 1), device=a.device, dtype=torch.float32)
    g = _lib.Wgrad()
    g.a, g.amax_a, g.b, g.amax_b, g.c, g.ws = _lib.ptr(a), _lib.ptr(amax_a), _lib.ptr(b), _lib.ptr(amax_b), _lib.ptr(out), _lib.ptr(ws)
    g.ws_floats, g.n_rows = n_ws, P
    g.lda, g.off_a, g.m, g.ldb, g.off_b, g.n, g.ldc, g.relu_b = lda, 0, m, ldb, 0, n_eff, ldc, int(bool(relu_b))
    g.colsum = _lib.ptr(cs)
    if gap_col is not None:
        g.b_gap_at, g.b_gap = gap_col, 1
        g.xcol, g.ld_xcol = b.data_ptr() + 4 * gap_col, ldb
        g.ccol, g.ld_ccol = out.data_ptr() + 4 * gap_col, ldc
    with _lib.on_device(a.device):
        rc = lib.e3dge_wgrad(ctypes.byref(g), _lib.stream_of(a))
    _lib.check(rc, "e3dge_wgrad")
    return (out, cs) if colsum else out
