"""GPU diagnostic for the packed decoder pipeline (csrc/decoder2.hip): stage-by-stage deviation from the planar (round-2) path
on the same GPU, final image vs the CPU oracle where that is cheap, per-launch HIP-event times, and the tile-shape sweep.

    python tools/dec2_check.py [--size 1024 --cm 2 --res 64 --batch 1] [--oracle] [--sweep] [--iters 20]

Prints one JSON line per measurement (also appended to gpurun_out/dec2_check.jsonl)."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import e3dge_amd  # noqa: E402,F401
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

DEV = "cuda:0"


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "dec2_check.jsonl"), "a") as f:
        f.write(line + "\n")


def planar_stages(dec, feats, latent, noise):
    """Intermediates of the planar path in the order of Decoder.dec2_unpack indices 1.. and the image."""
    os.environ["E3DGE_DECODER"] = "planar"
    try:
        from e3dge_amd import _lib
        outs = []
        n = dec.num_layers
        amax = torch.zeros((n + 1, _lib.AMAX_FLOATS), device=feats.device)
        mods = dec._all_modulations(latent)
        out = dec.conv1(feats, latent[:, 0], noise=noise[0], out_amax=amax[1], pre=mods[0])
        outs.append(out)
        skip = dec.to_rgb1(out, latent[:, 1], pre=mods[1])
        i, j = 1, 2
        for u in range(len(dec.to_rgbs)):
            out = dec.convs[2 * u](out, latent[:, i], noise=noise[2 * u + 1], in_amax=amax[i], out_amax=amax[i + 1], pre=mods[j])
            outs.append(out)
            out = dec.convs[2 * u + 1](out, latent[:, i + 1], noise=noise[2 * u + 2], in_amax=amax[i + 1], out_amax=amax[i + 2], pre=mods[j + 1])
            outs.append(out)
            skip = dec.to_rgbs[u](out, latent[:, i + 2], skip=skip, pre=mods[j + 2])
            i += 2
            j += 3
        return outs, skip
    finally:
        os.environ.pop("E3DGE_DECODER", None)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--cm", type=int, default=2)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--timing", action="store_true", help="-DE3DGE_PK_TIMING build (E3DGE_LIB_PATH): per-phase cycles of the convolutions")
    ap.add_argument("--scale", type=float, default=0.5, help="std of the synthetic feature map")
    a = ap.parse_args()
    torch.manual_seed(0)
    g = G_pred_latents(syn.model_opt(size=a.size, channel_multiplier=a.cm, renderer_spatial_output_dim=a.res),
                       syn.rendering_opt(N_samples=24), full_pipeline=True)
    syn.load_synthetic(g)
    sd = {k: v.clone() for k, v in g.state_dict().items()}
    g = g.to(DEV).eval()
    dec = g.decoder
    _, wd = syn.synthetic_inputs(a.batch, seed=1, device=DEV)
    wd = wd[:, :dec.n_latent].contiguous()
    feats = (a.scale * torch.randn(a.batch, dec.conv1.conv.in_channel, a.res, a.res, device=DEV)).contiguous()
    noise = [getattr(dec.noises, f"noise_{i}") for i in range(dec.num_layers)]
    tag = dict(size=a.size, cm=a.cm, res=a.res, batch=a.batch)
    with torch.no_grad():
        ref_stages, ref_img = planar_stages(dec, feats, wd, noise)
        ms = []
        img = dec._forward_packed(feats, wd, noise, kernel_ms=ms)
        torch.cuda.synchronize()
        errs = {}
        fused = os.environ.get("E3DGE_DEC2_FUSE_RGB", "1") != "0"
        for k, ref in enumerate(ref_stages[:-1] if fused else ref_stages):     # (the last activation is not stored when ToRGB is fused)
            got = dec.dec2_unpack(1 + k, feats.shape)
            errs[f"act{1 + k}"] = [float((got - ref).abs().max()), float(ref.abs().max())]
        got0 = dec.dec2_unpack(0, feats.shape)
        errs["act0"] = [float((got0 - feats).abs().max()), float(feats.abs().max())]
        errs["img"] = [float((img - ref_img).abs().max()), float(ref_img.abs().max())]
        emit(what="packed_vs_planar [max abs err, max abs ref]", **tag, **errs, finite=bool(torch.isfinite(img).all()))
        emit(what="kernel_ms", **tag, names=dec.dec2_launch_names(), ms=[round(x, 4) for x in ms], total=round(sum(ms), 4))
        if a.timing:
            st = dec._dec2_state(a.batch, a.res, torch.device(DEV))
            am = st['amax'].cpu()
            rows = {"conv1": 1}
            for u in range(len(dec.to_rgbs)):
                rows[f"L{u}.upblur(per TILE: wait, taps, H pass + LDS + barriers [gen 1: T->LDS], V pass + store [gen 1: blur + store])"] = 3 + 3 * u
                rows[f"L{u}.convT"] = 2 + 3 * u
                rows[f"L{u}.conv"] = 4 + 3 * u
            for name, r in rows.items():
                for wv, off in (("w0", 1), ("wl", 9)):
                    d = am[r, off:off + 6].tolist()
                    steps = max(d[5], 1)
                    emit(what="phase_cycles_per_step", layer=name, wave=wv, steps=int(steps), total=round(d[4] / steps), wait_barrier=round(d[0] / steps),
                         issue=round(d[1] / steps), mfma=round(d[2] / steps), epilogue=round(d[3] / steps))
                t4 = am[r, 17:21].tolist()
                emit(what="tap_cycles_total(w0)", layer=name, epi_step_tap0=round(t4[0]), epi_step_taps1_8=round(t4[1]), other_step_tap0=round(t4[2]), other_step_taps1_8=round(t4[3]))
        if a.oracle:
            from oracle import decoder_ref
            c = lambda t: t.detach().cpu()
            t0 = time.time()
            o32 = decoder_ref.decoder_forward(sd, c(feats), c(wd))
            o64 = decoder_ref.decoder_forward(sd, c(feats), c(wd), dtype=torch.float64)
            emit(what="vs_oracle", **tag, packed_vs_f32=float((c(img) - o32).abs().max()), packed_vs_f64=float((c(img).double() - o64).abs().max()),
                 planar_vs_f64=float((c(ref_img).double() - o64).abs().max()), f32_vs_f64=float((o32.double() - o64).abs().max()),
                 img_max=float(o64.abs().max()), oracle_s=round(time.time() - t0, 1))
        # end-to-end time of one forward, both paths
        t_packed = timed(lambda: dec._forward_packed(feats, wd, noise), a.iters)
        os.environ["E3DGE_DECODER"] = "planar"
        t_planar = timed(lambda: dec(feats, [wd], input_is_latent=True, randomize_noise=False), a.iters)
        os.environ.pop("E3DGE_DECODER")
        emit(what="forward_ms", **tag, packed=round(t_packed, 4), planar=round(t_planar, 4))
        # medians of the per-launch times over several forwards
        acc = []
        for _ in range(a.iters):
            m = []
            dec._forward_packed(feats, wd, noise, kernel_ms=m)
            acc.append(m)
        med = [round(sorted(col)[len(col) // 2], 4) for col in zip(*acc)]
        emit(what="kernel_ms_median", **tag, names=dec.dec2_launch_names(), ms=med, total=round(sum(med), 4))
        if a.sweep:
            names = dec.dec2_launch_names()
            for var, key in (("E3DGE_DEC2_S1", "conv"), ("E3DGE_DEC2_UP", "convT")):
                for v in range(5 if key == "conv" else 4):
                    os.environ[var] = str(v)
                    try:
                        acc = []
                        for _ in range(max(5, a.iters // 2)):
                            m = []
                            out = dec._forward_packed(feats, wd, noise, kernel_ms=m)
                            acc.append(m)
                        med = [sorted(col)[len(col) // 2] for col in zip(*acc)]
                        sel = {n: round(t, 4) for n, t in zip(names, med) if (key == "conv" and n.endswith("conv")) or (key == "conv" and n == "conv1") or (key == "convT" and n.endswith("convT"))}
                        emit(what="sweep", var=var, variant=v, **tag, ms=sel, img_err=float((out - ref_img).abs().max()))
                    except RuntimeError as e:
                        emit(what="sweep", var=var, variant=v, **tag, error=str(e)[:200])
                    finally:
                        os.environ.pop(var, None)


if __name__ == "__main__":
    main()
