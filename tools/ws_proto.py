"""GPU check + timing of the weight-stationary split-f16 layer chain (csrc/siren_ws.hip, e3dge_ws_chain) against float64.

    python tools/ws_proto.py [--points 98304] [--layers 8] [--iters 20]
"""
import argparse
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import e3dge_amd  # noqa: E402,F401
from e3dge_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=98304)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--grid", type=int, default=256)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda:0"
    L, P = a.layers, a.points
    W = (torch.rand(L, 256, 256) * 2 - 1) * (6.0 / 256) ** 0.5
    gamma = 1.0 + 0.5 * torch.rand(L, 256)
    beta = (torch.rand(L, 256) * 2 - 1) * 3.0
    x0 = torch.rand(P, 256) * 2 - 1
    film = torch.stack([gamma / 128.0, beta], 1).contiguous()                  # (L, 2, 256)
    if os.environ.get("WS_LIB"):                       # a variant built by tools/ws_variant.sh
        lib = ctypes.CDLL(os.environ["WS_LIB"])
        for name in ("e3dge_ws_image_bytes", "e3dge_ws_pack", "e3dge_ws_chain"):
            getattr(lib, name).restype, getattr(lib, name).argtypes = _lib.SIGNATURES[name]
    else:
        lib = _lib.load()
    fn = lib.e3dge_ws_chain
    d_film, d_x, d_w = film.to(dev), x0.to(dev), W.to(dev).contiguous()
    d_img = torch.empty(lib.e3dge_ws_image_bytes(L), dtype=torch.uint8, device=dev)
    assert lib.e3dge_ws_pack(d_img.data_ptr(), d_w.data_ptr(), L, torch.cuda.current_stream().cuda_stream) == 0
    y = torch.empty_like(d_x)
    dbg = torch.zeros(512, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = fn(d_img.data_ptr(), d_film.data_ptr(), d_x.data_ptr(), y.data_ptr(), L, P, a.grid, dbg.data_ptr(), st)
        assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc
    run()
    torch.cuda.synchronize()
    # float64 reference on a slice (and fp32 for scale)
    n_ref = min(P, 4096)
    xr = x0[:n_ref].double()
    x32 = x0[:n_ref].clone()
    for l in range(L):
        xr = torch.sin(gamma[l].double() * (xr @ W[l].double().T) + beta[l].double())
        x32 = torch.sin(gamma[l] * (x32 @ W[l].T) + beta[l])
    got = y[:n_ref].cpu().double()
    tail = y[-128:].cpu().double()
    xt = x0[-128:].double()
    for l in range(L):
        xt = torch.sin(gamma[l].double() * (xt @ W[l].double().T) + beta[l].double())
    res = dict(lib=os.path.basename(os.environ.get("WS_LIB", "default")), points=P, layers=L, err_vs_f64=float((got - xr).abs().max()), fp32_vs_f64=float((x32.double() - xr).abs().max()),
               tail_err=float((tail - xt).abs().max()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    flops = P * L * 256 * 256 * 2
    cyc = dbg[:256].cpu().double()
    res['shader_GHz'] = round(float(cyc.median() / dbg[256:].cpu().double().median()) * 0.1, 3)
    res['cycles_per_layer_group'] = round(float(cyc.median()) / max(L - 1, 1))
    res['cycles_per_point_layer'] = round(float(cyc.median()) / max(L - 1, 1) / 128, 1)
    res.update(ms=round(ms, 4), algorithmic_tflops=round(flops / ms / 1e9, 1), frac_of_f16_peak_over_3=round(flops / ms / 1e9 / 833.3, 3))
    print(json.dumps(res), flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "ws_proto.jsonl"), "a") as f:
        f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
