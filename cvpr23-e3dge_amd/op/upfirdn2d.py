"""upfirdn2d on the HIP kernel e3dge_upfirdn2d.

Public surface = project/models/op/upfirdn2d.py: `upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))` on
(B,C,H,W), differentiable to any order w.r.t. `input`.  Autograd is built on one fact (the same one the
reference's UpFirDn2dBackward uses, upfirdn2d.py:18-94,112-117): the adjoint of an upfirdn2d is another
upfirdn2d with the FIR flipped, up <-> down swapped and pads

    g_pad0 = k - pad0 - 1,      g_pad1 = in*up - out*down + pad0 - up + 1 ,

and the adjoint of that one is the original again -- so a single autograd.Function whose backward applies
itself to the adjoint geometry covers first, second, ... order.

`upfirdn2d_raw` keeps the argument list of the reference's pybind entry (upfirdn2d.cpp:12-23).

CPU tensors take a plain-PyTorch branch written here (zero-insertion, pad / crop, one F.conv2d with the flipped FIR,
strided slice -- the reference has a CPU branch too, upfirdn2d.py:146-200); ordinary torch autograd, never used for
GPU tensors."""
from typing import NamedTuple, Tuple

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib


class Geometry(NamedTuple):
    up: Tuple[int, int]            # (x, y)
    down: Tuple[int, int]          # (x, y)
    pad: Tuple[int, int, int, int]  # (x0, x1, y0, y1)

    def out_hw(self, in_h, in_w, kh, kw):
        (ux, uy), (dx, dy), (px0, px1, py0, py1) = self.up, self.down, self.pad
        return ((in_h * uy + py0 + py1 - kh) // dy + 1, (in_w * ux + px0 + px1 - kw) // dx + 1)

    def adjoint(self, in_h, in_w, kh, kw):
        """Geometry of the transposed operator, mapping (out_h, out_w) planes back to (in_h, in_w)."""
        (ux, uy), (dx, dy), (px0, _, py0, _) = self.up, self.down, self.pad
        out_h, out_w = self.out_hw(in_h, in_w, kh, kw)
        gx0, gy0 = kw - px0 - 1, kh - py0 - 1
        gx1 = in_w * ux - out_w * dx + px0 - ux + 1
        gy1 = in_h * uy - out_h * dy + py0 - uy + 1
        return Geometry((dx, dy), (ux, uy), (gx0, gx1, gy0, gy1))


def _launch(planes, kernel, geo):
    """planes (major, in_h, in_w) contiguous fp32, fp16 or fp64 on the GPU -> (major, out_h, out_w), same dtype.  The FIR taps go
    to the kernel as fp32 for fp32 / fp16 planes (the half form computes in fp32 and rounds once on store), as fp64 for fp64 planes."""
    major, in_h, in_w = planes.shape
    kernel = kernel.double().contiguous() if planes.dtype == torch.float64 else kernel.float()
    kh, kw = kernel.shape
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = geo.up, geo.down, geo.pad
    lib = _lib.load()
    out_h = lib.e3dge_upfirdn2d_out_size(in_h, uy, dy, py0, py1, kh)
    out_w = lib.e3dge_upfirdn2d_out_size(in_w, ux, dx, px0, px1, kw)
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError(f"upfirdn2d: empty output for input {in_h}x{in_w}, up {geo.up}, down {geo.down}, "
                           f"pad {geo.pad}, kernel {kh}x{kw}")
    y = planes.new_empty((major, out_h, out_w))
    fn = lib.e3dge_upfirdn2d_f16 if planes.dtype == torch.float16 else (lib.e3dge_upfirdn2d_f64 if planes.dtype == torch.float64 else lib.e3dge_upfirdn2d)
    with torch.cuda.device(planes.device):
        rc = fn(_lib.ptr(y), _lib.ptr(planes), _lib.ptr(kernel), major, in_h, in_w, kh, kw,
                ux, uy, dx, dy, px0, px1, py0, py1, _lib.stream_of(planes))
    _lib.check(rc, "e3dge_upfirdn2d")
    return y


def upfirdn2d_raw(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """input (major, in_h, in_w, 1) -> (major, out_h, out_w, 1); minor_dim must be 1 (as every reference
    caller passes it, upfirdn2d.py:27,78,96)."""
    _lib.require_gpu(input, "input", half_ok=True)
    _lib.require_gpu(kernel, "kernel", half_ok=True)
    if input.ndim != 4 or input.shape[3] != 1:
        raise RuntimeError("upfirdn2d_raw expects a (major, in_h, in_w, 1) tensor")
    geo = Geometry((up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1))
    return _launch(input.contiguous().squeeze(3), kernel.contiguous(), geo).unsqueeze(3)


class _UpFirDn(Function):
    @staticmethod
    def forward(ctx, x, kernel, geo, out_hw):
        B, C, in_h, in_w = x.shape
        y = _launch(x.contiguous().reshape(B * C, in_h, in_w), kernel.contiguous(), geo)
        if out_hw is not None and tuple(y.shape[1:]) != tuple(out_hw):
            raise RuntimeError(f"upfirdn2d adjoint produced {tuple(y.shape[1:])}, expected {tuple(out_hw)}")
        ctx.save_for_backward(kernel)
        ctx.geo, ctx.in_hw = geo, (in_h, in_w)
        return y.reshape(B, C, y.shape[1], y.shape[2])

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        kh, kw = kernel.shape
        adj = ctx.geo.adjoint(ctx.in_hw[0], ctx.in_hw[1], kh, kw)
        gx = _UpFirDn.apply(gy, torch.flip(kernel, [0, 1]), adj, ctx.in_hw)
        return gx, None, None, None


def _upfirdn2d_cpu(x, kernel, geo):
    """(B,C,H,W) CPU tensor: up-sample by zero insertion, pad (negative = crop), correlate with the flipped FIR, decimate."""
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = geo.up, geo.down, geo.pad
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    v = x.reshape(B * C, 1, H, 1, W, 1)
    v = F.pad(v, [0, ux - 1, 0, 0, 0, uy - 1]).reshape(B * C, 1, H * uy, W * ux)
    v = F.pad(v, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    v = v[:, :, max(-py0, 0):v.shape[2] - max(-py1, 0), max(-px0, 0):v.shape[3] - max(-px1, 0)]
    if v.shape[2] < kh or v.shape[3] < kw:
        raise RuntimeError(f"upfirdn2d: empty output for input {H}x{W}, up {geo.up}, down {geo.down}, pad {geo.pad}, "
                           f"kernel {kh}x{kw}")
    v = F.conv2d(v, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(v.dtype))
    v = v[:, :, ::dy, ::dx]
    return v.reshape(B, C, v.shape[2], v.shape[3])


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if isinstance(input, torch.Tensor) and input.device.type == "cpu":
        if input.ndim != 4 or kernel.ndim != 2:
            raise RuntimeError("upfirdn2d expects input (B, C, H, W) and a 2-D FIR kernel")
        return _upfirdn2d_cpu(input, kernel, Geometry((up, up), (down, down), (pad[0], pad[1], pad[0], pad[1])))
    _lib.require_gpu(input, "input", half_ok=True)
    _lib.require_gpu(kernel, "kernel", half_ok=True)
    if input.ndim != 4 or kernel.ndim != 2:
        raise RuntimeError("upfirdn2d expects input (B, C, H, W) and a 2-D FIR kernel")
    geo = Geometry((up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
    return _UpFirDn.apply(input, kernel, geo, None)
