#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric on MI355X: rendered rays/s at 64x64 rays x 24 samples.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts the N ranks itself (one process per GPU through torch.distributed.run, rendezvous on
127.0.0.1) when it was not already launched by torchrun, and fails loudly if fewer than N GPUs are visible; when the driver
launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) RANK / LOCAL_RANK / WORLD_SIZE
come from the environment as usual.

One step = one pass of the hot path over one batch of synthetic input on every rank: W+ codes -> FiLM parameters
(e3dge_film_params) -> fused ray generation + SIREN + compositing (e3dge_siren_render_fwd) for ONE 64x64 image
x 24 samples per GPU (BASELINE.json configs[1]; inputs already resident in HBM).  Images shard across ranks
with no data-path collective (weak scaling); `value` = rays rendered by all ranks / max-over-ranks time.

Before the W warm-up steps the same launches run untimed for --prewarm-ms (default 250 ms, reported as `prewarm_ms`): the
GPU clocks down during the seconds of host-side setup and the first ~50-100 ms of load run slower; W and K are unchanged.

Extra objects on the JSON line:
  roofline      dominant kernel of the headline (default) mode: algorithmic FLOPs per launch / its mean duration measured
                with HIP events on the launch stream inside the timed region.
  modes         the same K-step measurement for BOTH contraction modes, f16x3 (default) and strict f32.
  sustained     the K-step block repeated for >= 1 s: median and spread of rays/s (the K-step line alone is ~0.1 s).
  c3            BASELINE configs[2]: per-image evaluation (pass #1, texture head, pass #2, decoder 64^2 -> 1024^2, 8 metric
                scalars) sharded over the ranks with one all_gather of the metric rows.
  c4            BASELINE configs[3]: the 120-pose sweep at 128x128x48, sequential and batched 8 poses per launch.
  surface       surface extraction, device half: 128x128 rays x 128 samples + align_volume.
  train_step_ms BASELINE configs[4], renderer part: stage-1 step at 64x64x18 with the eikonal losses, fwd + bwd.
  inversion_fwd_ms  pass #1 + texture head + pass #2 + decoder to 1024^2, one image (inversion_fwd_graph_ms: the same launches
                replayed as one HIP graph).
  train_step    (all ranks) the same step with the encoder's 1.03 GB fp32 gradient all-reduce (trainer.py:1737-1778 through DDP,
                dist_utils.py:108-130) emulated on a side stream: allreduce_ms, overlapped ms, overlap_frac.
  inversion     per-kernel table of the inversion forward (HIP events): ms, bound, achieved / peak, frac for both render passes, the
                texture head and every launch of the decoder.
  autograd      forward + backward of the decoder (1024^2) and of Fuse_sft_MLP (98,304 points) with a graph wanted -- inputs require grad,
                parameters frozen, the shape of a train_ae.py forward: autograd nodes with the native forward vs the library path.
  cpu_baseline  the oracle restatement (oracle/renderer_ref.py, "port": bit-identical to the reference's PyTorch path on
                the golden vectors) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

RES, N_SAMPLES = 64, 24
MAC_PER_POINT = 3 * 256 + 7 * 256 * 256 + 259 * 256 + 256 * 3 + 256       # 526,848
FLOP_PER_RAY = 2 * MAC_PER_POINT * N_SAMPLES                               # 25.29 MFLOP
BYTES_PER_RAY = (264 + 5 * N_SAMPLES) * 4                                  # mandatory outputs, 1,536 B
PEAK_F32_MFMA_TFLOPS = 157.3                                               # MI355X_MICROARCH.md (dense fp32 MFMA)
PEAK_F16_MFMA_TFLOPS = 2500.0                                              # dense f16 / bf16 MFMA (NOT the 2:1-sparse figure)
TRAFFIC_FILE = os.path.join(REPO, "profiles", "traffic_pmc.json")          # written by tools/refresh_traffic.sh (PMC passes)
PEAK_HBM_GBPS = 8000.0                                                     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s copy ceiling)
C3_IMAGES = 2824                                                           # project/utils/setup/train_setup.py:332
ENCODER_GRAD_BYTES = 1.03e9                                                # fp32 gradient volume of the encoder per step


SHORT_NOTES = {      # the prose lives in DESIGN.md section 5 ("fields of the bench line"); the line carries <= 120 characters per note
    "roofline/traffic_note": "PMC FETCH_SIZE x2 + WRITE_SIZE per launch, profiles/traffic_pmc.json (tools/refresh_traffic.sh); {stale}",
    "c3/note": "configs[2]: pass #1, texture head + FiLM, pass #2, decoder to 1024^2, 8 metrics per image; 2,824 images sharded i mod W",
    "local_features/note": "que_render_given_ref's per-point features: 3 gathers + Fuse_sft_MLP (9 launches) + positional encoding",
    "inversion_fwd_note": "pass #1 + texture head/FiLM + pass #2 on the layer-7 record + decoder 64^2->1024^2; encoder / image filters excluded",
    "inversion/note": "HIP-event ms per launch group (median of 10); frac = max(flops / (f16 peak / 3), bytes / 8 TB/s) / measured time",
    "c4/note": "configs[3]: 120 poses of one latent, 128x128 rays x 48 samples, one launch per pose, pose k -> rank k mod W",
    "surface/note": "surf_extraction generator: 128x128 rays x 128 samples + align_volume onto the 128^3 grid",
    "train_step_note": "C5 renderer part, 64x64x18 per GPU: forward + eikonal + surface normals, backward to the styles incl. double backward",
    "train_step/roofline/note": "bound = the 30 GEMM chains of the step on f16 MFMA / 3; saved-state bytes are a design cost, not algorithmic",
    "cpu_baseline/sample": "{best}",
}


def finalize(result):
    """Compact form of the line the driver records (it keeps the contract keys, `roofline`, `cpu_baseline` and the last KBs of
    stdout): notes <= 120 characters, floats to 6 significant digits, the per-launch table as rows, and a `summary` of the
    secondary figures LAST so that it sits in the recorded tail."""
    keep_exact = {"value", "ms_per_step"}

    def shorten(path, v):
        if not isinstance(v, str) or len(v) <= 120:
            return v
        t = SHORT_NOTES.get(path)
        if t is None:
            return v[:117] + "..."
        stale = "STALE (kernel sources changed)" if "STALE" in v else ("unchanged sources" if "unchanged since" in v else "")
        best = v.split(" (oracle")[0][-110:] if path == "cpu_baseline/sample" else ""
        return t.format(stale=stale, best=best)[:120]

    def walk(o, path):
        if isinstance(o, dict):
            return {k: walk(v, (path + "/" + k) if path else k) for k, v in o.items()}
        if isinstance(o, list):
            return [walk(v, path + "[]") for v in o]
        if isinstance(o, float) and path not in keep_exact:
            return float(f"{o:.6g}")
        return shorten(path, o)
    r = walk(result, "")
    inv = r.get("inversion")
    if isinstance(inv, dict) and isinstance(inv.get("kernels"), list):
        rows = []
        for k in inv["kernels"]:
            rows.append([k.get("name", "")[:46], round(k.get("ms", 0.0), 4), k.get("bound", ""), round(k.get("frac", 0.0), 3)])
        inv["kernels"] = rows
        inv["kernel_cols"] = ["name", "ms", "bound", "frac"]
    g = lambda *ks: __import__("functools").reduce(lambda d, k: d.get(k) if isinstance(d, dict) else None, ks, r)
    summary = {"rays_per_s": r.get("value"), "roofline_frac": g("roofline", "frac"), "sustained_median_rays_per_s": g("sustained", "median_rays_per_s"),
               "strict_f32_rays_per_s": r.get("strict_f32_rays_per_s"), "strict_f32_frac": r.get("strict_f32_frac"),
               "inversion_fwd_ms": r.get("inversion_fwd_ms"), "inversion_fwd_graph_ms": r.get("inversion_fwd_graph_ms"),
               "inversion_fwd_graph_steady_ms": r.get("inversion_fwd_graph_steady_ms"),
               "inversion_fwd_no_reuse_ms": r.get("inversion_fwd_no_reuse_ms"), "c3_images_per_s": g("c3", "images_per_s"),
               "ranks_joined": r.get("ranks_joined"), "c3_per_rank_images_per_s_min": g("c3", "per_rank_images_per_s_min"),
               "c3_per_rank_images_per_s_max": g("c3", "per_rank_images_per_s_max"),
               "train_step_allreduce_busbw_GBps": g("train_step", "allreduce_busbw_GBps"),
               "train_step_with_allreduce_ms": g("train_step", "step_with_allreduce_ms"),
               "c4_ms_per_pose": g("c4", "sequential", "ms_per_pose"), "train_step_ms": r.get("train_step_ms"),
               "train_step_full_ms": r.get("train_step_full_ms"), "train_step_stage2_ms": r.get("train_step_stage2_ms"), "train_step_full_both_latents_ms": r.get("train_step_full_both_latents_ms"), "train_step_full_batch4_ms_per_sample": r.get("train_step_full_batch4_ms_per_sample"), "train_step_batch4_ms_per_sample": r.get("train_step_batch4_ms_per_sample"), "decoder_packed_fwd_bwd_ms": g("autograd", "decoder_packed_fwd_bwd_ms"),
               "decoder_library_fwd_bwd_ms": g("autograd", "decoder_library_fwd_bwd_ms"),
               "fuse_sft_hip_fwd_bwd_ms": g("autograd", "fuse_sft_hip_fwd_bwd_ms"), "tex_head_fwd_bwd_ms": g("autograd", "tex_head_fwd_bwd_ms"),
               "blur_hbm_frac_1024": g("stream_ops", "blur_f32", "hbm_frac"), "bias_act_hbm_frac_1024": g("stream_ops", "bias_act_f32", "hbm_frac"),
               "blur_f16_hbm_frac_1024": g("stream_ops", "blur_f16", "hbm_frac"), "bias_act_f16_hbm_frac_1024": g("stream_ops", "bias_act_f16", "hbm_frac"),
               "train_step_mfma_frac": g("train_step", "roofline", "frac"), "train_step_f32_fallback_ms": r.get("train_step_f32_fallback_ms"), "cpu_rays_per_s": g("cpu_baseline", "value")}
    if isinstance(inv, dict) and isinstance(inv.get("kernels"), list):
        dec = sum(k[1] for k in inv["kernels"] if k[0].startswith("decoder:"))
        summary["decoder_ms_sum_of_launches"] = round(dec, 4)
    # the prose of these objects is DESIGN.md section 5's; their notes and the stream_ops detail are dropped from the line (its four fractions are in
    # `summary`) so that the whole line stays within the tail the driver records
    for key in ("autograd", "c3", "local_features", "c4", "surface", "inversion", "train_step_stage2", "train_step_full"):
        if isinstance(r.get(key), dict):
            r[key].pop("note", None)
            if isinstance(r[key].get("roofline"), dict):
                r[key]["roofline"].pop("note", None)
    if isinstance(g("autograd", "decoder_packed_fwd_bwd_roofline"), dict):
        r["autograd"]["decoder_packed_fwd_bwd_roofline"].pop("note", None)
    r.pop("stream_ops", None)
    for key in ("train_step_full_note", "train_step_note", "inversion_fwd_note"):       # (DESIGN.md section 5 says what these legs are)
        r.pop(key, None)
    r["summary"] = summary
    return r


def kernel_source_digest():
    """Digest of the render-kernel sources: stamps profiles/traffic_pmc.json so that a stale PMC figure is detectable."""
    import hashlib
    h = hashlib.sha256()
    for name in ("siren.hip", "siren16.h", "siren_common.h"):
        with open(os.path.join(REPO, "cvpr23-e3dge_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pin_rank(local_rank, world):
    """One contiguous core set per rank (8 Python launchers on a 256-core host would otherwise share cores and oversubscribe the
    BLAS / OpenMP pools); returns (cores, threads)."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None, None
    per = max(1, len(avail) // max(world, 1))
    mine = avail[local_rank * per:(local_rank + 1) * per] or avail
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None, None
    nt = max(1, min(8, len(mine)))
    os.environ.setdefault("OMP_NUM_THREADS", str(nt))
    return len(mine), nt


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (metric config: 1)")
    ap.add_argument("--c3-images", type=int, default=0,
                    help="images per rank in the C3 leg (0 = the whole evaluation set split over the ranks: ceil(2824 / N))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inversion", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-c4", action="store_true")
    ap.add_argument("--no-surface", action="store_true")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-modes", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the K-step headline measurement (profiling runs)")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed load before the W warm-up steps (and before every informational leg): the GPU clocks down "
                         "during the seconds of host-side setup and needs ~50-100 ms of load to come back")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only (gloo, no GPU work): prints the number of ranks that joined")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks here.  Returns the children's exit code."""
    if not args.dry_run:
        import torch
        n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_vis < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["E3DGE_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_inversion = args.no_train_step = args.no_c4 = args.no_c3 = args.no_surface = True
        args.no_modes = args.no_sustained = True
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let bench.py start them)")

    cores_pinned, host_threads = pin_rank(local_rank, world)
    import torch
    if host_threads:
        torch.set_num_threads(host_threads)
    c3_per_rank = args.c3_images if args.c3_images > 0 else -(-C3_IMAGES // world)
    dist = None
    if args.dry_run:
        # launcher / rendezvous / work-split check without a GPU: every rank reports its shard of the legs, rank 0 prints the plan
        import torch.distributed as dist
        from e3dge_amd import sharded_eval as se
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        mine = torch.tensor([len(se.shard_indices(c3_per_rank * world, rank, world)), len(se.shard_indices(120, rank, world)),
                             cores_pinned or 0], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        g = torch.zeros(1024)                 # stands in for the 1.03 GB gradient bucket
        dist.all_reduce(g)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_joined": int(t.item()),
                              "self_launched": os.environ.get("E3DGE_BENCH_SELF_LAUNCHED") == "1",
                              "c3": {"images": c3_per_rank * world, "images_per_rank": [int(a[0]) for a in allr]},
                              "c4": {"poses": 120, "poses_per_rank": [int(a[1]) for a in allr]},
                              "train_step": {"allreduce_bytes": ENCODER_GRAD_BYTES, "ranks": world},
                              "cores_per_rank": [int(a[2]) for a in allr], "host_threads": host_threads}))
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_joined = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        ranks_joined = int(t.item())
        if ranks_joined != world:
            raise SystemExit(f"{ranks_joined} ranks joined the RCCL group, expected {world}")

    import e3dge_amd  # noqa: F401
    from e3dge_amd import sharded_eval, synthetic as syn
    from e3dge_amd.camera_utils import generate_camera_params
    from e3dge_amd.stylesdf_model import G_pred_latents
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer

    g = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=N_SAMPLES), full_pipeline=True)
    syn.load_synthetic(g)
    sd_cpu = {k: v.clone() for k, v in g.state_dict().items()}
    g = g.to(dev).eval()
    g.requires_grad_(False)
    renderer = g.renderer
    B = args.batch
    wr, wd = syn.synthetic_inputs(B, seed=1 + 17 * rank, device=dev)      # every rank renders its own image(s)
    poses, focal, near, far, _ = generate_camera_params(RES, dev, locations=torch.zeros(B, 2, device=dev))
    renderer.siren.device_image()                                          # weight image packed once, outside the loop
    default_mode = renderer.siren.mfma_mode

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        """[x of rank 0, ..., x of rank W-1]"""
        if dist is None:
            return [x]
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def timed_block(steps, with_events):
        """K steps between barriers; returns (elapsed_s max over ranks, mean kernel ms or None, last output)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if with_events else None
        barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            film = renderer.siren.film_params(wr)
            if ev:
                ev[i][0].record()                                          # same stream the kernel is launched on
            out = renderer.render_with_film(film, focal, poses, near, far)
            if ev:
                ev[i][1].record()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                                     # this rank's own K steps (before the closing barrier)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        timed_block.own_s = own
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(steps, 1) if ev else None
        return elapsed, kern_ms, out

    def spin(ms):
        """Untimed: the headline launches back to back for `ms` milliseconds (brings the clocks up after host-side idling)."""
        if ms <= 0:
            return
        with torch.no_grad():
            t_end = time.perf_counter() + ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(20):
                    renderer.render_with_film(renderer.siren.film_params(wr), focal, poses, near, far)
                torch.cuda.synchronize()

    def roofline_of(mode, kern_ms):
        flops = FLOP_PER_RAY * B * RES * RES                    # ALGORITHMIC flops (one fp32 multiply-add per weight per point)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        if mode == "f32":
            peak, note = PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA peak"
        else:   # every algorithmic product costs three f16 MFMA products
            peak, note = PEAK_F16_MFMA_TFLOPS / 3.0, "dense f16 MFMA peak / 3 (the split needs 3 f16 products per fp32-accurate product)"
        traffic, tnote = None, f"no {os.path.relpath(TRAFFIC_FILE, REPO)}"
        extra_pmc = {}
        try:
            with open(TRAFFIC_FILE) as f:
                tj = json.load(f)
            ent = tj.get(mode)
            if ent and B == 1:
                traffic = int(2 * ent["FETCH_SIZE_KB"] * 1024 + ent["WRITE_SIZE_KB"] * 1024)
                if "mfma_busy_frac" in ent:
                    extra_pmc = {"mfma_busy_frac": ent["mfma_busy_frac"], "shader_clock_ghz": ent.get("shader_clock_ghz"),
                                 "mfma_insts_per_launch": ent.get("SQ_INSTS_MFMA")}
                stale = tj.get("kernel_source_digest") != kernel_source_digest()
                tnote = (f"NOT measured in this run: PMC FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch from "
                         f"{os.path.relpath(TRAFFIC_FILE, REPO)} (tools/refresh_traffic.sh, separate rocprofv3 --pmc passes; taken at git "
                         f"{tj.get('git', '?')}, kernel sources {'CHANGED since: STALE' if stale else 'unchanged since'})")
        except (OSError, ValueError, KeyError):
            pass
        kname = {"f32": "siren_kernel<0,0,false>", "f16x3": "siren16_kernel<0,false>", "f16x3_v1": "siren_kernel<0,1,false>"}[mode]
        return {"bound": "mfma", "kernel": f"{kname} (e3dge_siren_render_fwd, {mode})",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": note,
                "achieved_over_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS, "kernel_ms": kern_ms, "flop_per_launch": flops,
                "algorithmic_output_bytes_per_launch": BYTES_PER_RAY * B * RES * RES,
                "hbm_frac_of_8TBps": BYTES_PER_RAY * B * RES * RES / (kern_ms * 1e-3) / 8e12,
                "traffic": traffic, "traffic_note": tnote, **extra_pmc}

    rays_per_step = B * RES * RES * world
    dtype_of = lambda m: "f32" if m == "f32" else "f32 (operands split f16 hi+lo, 3 f16 MFMA products, fp32 accumulate)"

    # ---------------------------------------------------------------- headline: W warm-up steps, exactly K timed steps
    spin(args.prewarm_ms)
    with torch.no_grad():
        for _ in range(args.warmup):
            renderer.render_with_film(renderer.siren.film_params(wr), focal, poses, near, far)
        elapsed, kern_ms, out = timed_block(args.steps, True)
    assert torch.isfinite(out['gen_thumb_imgs']).all()
    per_rank_ms = [1e3 * x / args.steps for x in all_ranks(timed_block.own_s)]
    value = rays_per_step * args.steps / elapsed
    result = {
        "metric": "rendered_rays_per_sec_64x64x24", "value": value, "unit": "rays/s", "n_gpus": world, "ranks_joined": ranks_joined,
        "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms, "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_per_rank": {"min": min(per_rank_ms), "max": max(per_rank_ms)}, "cores_per_rank": cores_pinned,
        "host_threads_per_rank": host_threads,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_of(default_mode), "data": "synthetic",
        "config": {"workload": "C2: single-image W+ -> volume render, 64x64 rays x 24 samples per ray, "
                               f"{B} image(s) per GPU per step (film_params + fused render launch)",
                   "rays_per_gpu_per_step": B * RES * RES, "samples_per_ray": N_SAMPLES, "parallelism": f"images sharded x{world}",
                   "mfma_mode": default_mode},
    }
    if rank == 0:
        result["roofline"] = roofline_of(default_mode, kern_ms)

    # ---------------------------------------------------------------- sustained: repeat the K-step block for >= 1 s
    if not args.no_sustained:
        blocks, t_start = [], time.perf_counter()
        with torch.no_grad():
            while True:
                e, _, _ = timed_block(args.steps, False)
                blocks.append(rays_per_step * args.steps / e)
                done = max_over_ranks(time.perf_counter() - t_start)       # same decision on every rank
                if done >= 1.0 and len(blocks) >= 3 or len(blocks) >= 200:
                    break
        result["sustained"] = {"blocks": len(blocks), "steps_per_block": args.steps, "seconds": done,
                               "median_rays_per_s": statistics.median(blocks), "min_rays_per_s": min(blocks),
                               "max_rays_per_s": max(blocks),
                               "spread_frac": (max(blocks) - min(blocks)) / statistics.median(blocks)}

    # ---------------------------------------------------------------- both contraction modes, same K-step measurement
    if not args.no_modes:
        modes = {}
        for mode in ("f16x3", "f32"):
            if mode == default_mode:
                e_m, k_m = elapsed, kern_ms
            else:
                renderer.siren.mfma_mode = mode
                with torch.no_grad():
                    for _ in range(max(3, args.warmup // 4)):
                        renderer.render_with_film(renderer.siren.film_params(wr), focal, poses, near, far)
                    e_m, k_m, _ = timed_block(args.steps, True)
                renderer.siren.mfma_mode = default_mode
            rf = roofline_of(mode, k_m)
            modes[mode] = {"value": rays_per_step * args.steps / e_m, "unit": "rays/s", "ms_per_step": 1e3 * e_m / args.steps,
                           "kernel_ms": k_m, "dtype": dtype_of(mode), "achieved_tflops": rf["achieved"], "peak_tflops": rf["peak"],
                           "frac": rf["frac"], "traffic": rf["traffic"]}
        result["modes"] = modes
        result["strict_f32_rays_per_s"] = modes["f32"]["value"]
        result["strict_f32_frac"] = modes["f32"]["frac"]

    # ---------------------------------------------------------------- C3: per-image evaluation sharded over the ranks
    gl = None
    need_local = (not args.no_c3) or (rank == 0 and not args.no_inversion)
    if need_local:
        gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=N_SAMPLES, enable_local_model=True,
                                                               L_pred_tex_modulations=True), full_pipeline=True)
        sd_l = {k.replace('renderer.network.', 'renderer.network.netGlobal.'): v for k, v in sd_cpu.items()}
        for k, v in gl.state_dict().items():
            if '.netLocal.' in k:                 # texture head: small synthetic weights (the reference zero-inits it)
                sd_l[k] = 0.05 * syn.synthetic_tensor(k, v.shape)
        gl.load_state_dict(sd_l)
        gl = gl.to(dev).eval()
        gl.requires_grad_(False)
        p1, f1, n1, fa1, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
        feats = syn.synthetic_local_feats(1, RES, N_SAMPLES, device=dev)      # what the PIFu branch would deliver
        target = torch.tanh(torch.randn(1, 3, 1024, 1024, device=dev, generator=torch.Generator(device=dev).manual_seed(7)))

        def inversion(w_r, w_d):
            gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)       # pass #1
            return gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False,
                      local_data_batch={'feats': feats})                    # tex head + pass #2 + decoder

    def inversion_kernel_table(gl_, w_r, w_d, out2):
        """[{name, ms, bound, achieved, peak, unit, frac}]: both render passes, the texture head, every launch of the decoder."""
        def ev_ms(fn, n=10, rounds=5):
            """ms per call: n calls back to back between one pair of events (a lone call after a sync measures the idle GPU's
            launch latency and clock ramp, not the kernel), median over `rounds`"""
            fn()
            ts = []
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / n)
            return statistics.median(ts)

        def row(name, ms, flops=0.0, nbytes=0.0):
            t_f = flops / (PEAK_F16_MFMA_TFLOPS / 3.0 * 1e12)
            t_b = nbytes / (PEAK_HBM_GBPS * 1e9)
            if ms <= 0:
                return {"name": name, "ms": 0.0, "note": "fused into the previous launch"}
            if t_f >= t_b:
                return {"name": name, "ms": ms, "bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
                        "unit": "TFLOP/s", "frac": t_f / (ms * 1e-3), "hbm_GBps": nbytes / (ms * 1e-3) / 1e9}
            return {"name": name, "ms": ms, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": t_b / (ms * 1e-3), "mfma_TFLOPs": flops / (ms * 1e-3) / 1e12}
        rr = gl_.renderer
        film = rr.siren.film_params(w_r)
        head = rr.network.netLocal.local_feat_to_tex_modulations_linear
        tex = head.tex_modulations(feats)
        n_pts = RES * RES * N_SAMPLES
        reuse = rr._reuse_enabled(None)
        key = rr._reuse_key(w_r, f1, p1, n1, fa1) if reuse else None
        view_flop = 2.0 * (259 * 256 + 3 * 256) * N_SAMPLES          # view layer + rgb head per ray
        rows = [row("render pass #1 (siren16_kernel" + (", + layer-7 record for pass #2)" if reuse else ")"),
                    ev_ms(lambda: rr.render_with_film(film, f1, p1, n1, fa1, reuse_key=key)),
                    flops=FLOP_PER_RAY * RES * RES, nbytes=(BYTES_PER_RAY + (1024 * N_SAMPLES if reuse else 0)) * RES * RES),
                row("texture head alone (resblock_kernel, 301 -> 512, (alpha, beta) to HBM)", ev_ms(lambda: head.tex_modulations(feats)),
                    flops=2 * 398825 * n_pts, nbytes=n_pts * (301 + 512) * 4)]
        if reuse:       # (the record written by the timed pass-#1 launches above is the one these read)
            from e3dge_amd import volume_renderer as _vr
            fused = _vr._fuse_texfilm() and rr._lazy_tex_ok(feats, head)
            if fused:   # what the forward runs: head + FiLM in one launch (FiLM-ed record out), then the short pass #2 on that record
                rec = _vr._BACKBONE.get(rr)
                tbuf = torch.zeros_like(rec['buf'])
                rows.append(row("texture head + FiLM (resblock_kernel<FILM>: 301 -> 512, h' = (alpha+1) h8 + beta -> record; (alpha, beta) stay on chip)",
                                ev_ms(lambda: head.tex_film(feats, rec['buf'], tbuf, 1, RES, RES, N_SAMPLES)),
                                flops=2 * 398825 * n_pts, nbytes=n_pts * (301 * 4 + 1024 + 1024)))
                rows.append(row("render pass #2: view layer + compositing on the FiLM-ed layer-7 record",
                                ev_ms(lambda: rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=_vr._LazyTex(head, feats), reuse_key=key)) - rows[-1]["ms"],
                                flops=view_flop * RES * RES, nbytes=(BYTES_PER_RAY + (1024 + 4) * N_SAMPLES) * RES * RES))
                rows[-1]["note"] = "timed as (head + FiLM + pass #2) minus the head + FiLM row"
            rows.append(row("render pass #2 on record + (alpha, beta) from HBM (E3DGE_FUSE_TEXFILM=0" + ("; not part of the forward)" if fused else ")"),
                            ev_ms(lambda: rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=tex, reuse_key=key)),
                            flops=view_flop * RES * RES, nbytes=(BYTES_PER_RAY + (1024 + 4 + 2 * 256 * 4) * N_SAMPLES) * RES * RES))
        rows.append(row("render pass #2 as a full launch (E3DGE_REUSE_BACKBONE=0" + ("; not part of the forward)" if reuse else ")"),
                        ev_ms(lambda: rr.render_with_film(film, f1, p1, n1, fa1, tex_conditions=tex)),
                        flops=FLOP_PER_RAY * RES * RES, nbytes=(BYTES_PER_RAY + 2 * 256 * 4 * N_SAMPLES) * RES * RES))
        dec = gl_.decoder
        latent, noise = dec.styles_and_noise_forward([w_d], None, input_is_latent=True, randomize_noise=False)
        fmap = out2['features'].contiguous()
        if not dec._dec2_ok(fmap, latent, noise, None):
            return rows + [{"name": "decoder", "note": "packed pipeline not taken (E3DGE_DECODER=planar or unsupported shape)"}]
        acc = []
        for _ in range(10):
            ms = []
            dec._forward_packed(fmap, latent, noise, kernel_ms=ms)
            acc.append(ms)
        med = [statistics.median(c) for c in zip(*acc)]
        names = dec.dec2_launch_names()
        convs = [dec.conv1.conv] + [c.conv for c in dec.convs]
        r = fmap.shape[2]
        sizes = {}
        wbytes = sum(c.in_channel * c.out_channel * 9 for c in convs) * 8           # wpre read + hi/lo image written
        c0 = convs[0]
        sizes["styles"] = (0.0, sum(m.in_channel for m, _ in dec._mod_layers()) * dec.style_dim * 4.0)
        sizes["amax(features)"] = (0.0, 4.0 * fmap.numel())
        sizes["pack(features)"] = (0.0, 8.0 * fmap.numel())
        sizes["weights"] = (0.0, float(wbytes))
        sizes["conv1"] = (2.0 * 9 * c0.in_channel * c0.out_channel * r * r, 4.0 * (c0.in_channel + c0.out_channel) * r * r)
        sizes["to_rgb1"] = (0.0, 4.0 * c0.out_channel * r * r)
        for u in range(len(dec.to_rgbs)):
            cu, cc = convs[1 + 2 * u], convs[2 + 2 * u]
            t_b = 4.0 * cu.out_channel * (2 * r + 1) ** 2
            sizes[f"L{u}.convT"] = (2.0 * 9 * cu.in_channel * cu.out_channel * r * r, 4.0 * cu.in_channel * r * r + t_b)
            r *= 2
            sizes[f"L{u}.blur"] = (0.0, t_b + 4.0 * cu.out_channel * r * r)
            last = u == len(dec.to_rgbs) - 1
            sizes[f"L{u}.conv"] = (2.0 * 9 * cc.in_channel * cc.out_channel * r * r, 4.0 * (cc.in_channel + cc.out_channel) * r * r)
            sizes[f"L{u}.to_rgb"] = (0.0, 4.0 * cc.out_channel * r * r + 12.0 * r * r)
            if med[names.index(f"L{u}.blur")] == 0.0:                      # blur folded into the transposed conv: T never exists
                sizes[f"L{u}.convT"] = (sizes[f"L{u}.convT"][0], 4.0 * cu.in_channel * (r // 2) ** 2 + 4.0 * cu.out_channel * r * r)
            if last and med[names.index(f"L{u}.to_rgb")] == 0.0:           # ToRGB folded into the conv: the activation is never stored
                sizes[f"L{u}.conv"] = (sizes[f"L{u}.conv"][0], 4.0 * cc.in_channel * r * r + 12.0 * r * r)
        return rows + [row("decoder: " + n, t, *sizes[n]) for n, t in zip(names, med)]

    if not args.no_c3:
        try:
            n_units = c3_per_rank * world
            # (64 distinct latent pairs cycled over the set: building 2,824 synthetic codes is host time, not the workload)
            pool = [syn.synthetic_inputs(1, seed=1000 + i, device=dev) for i in range(64)]

            def unit(i):
                w_r, w_d = pool[(i * 7 + rank) % 64]
                o = inversion(w_r, w_d)
                return sharded_eval.image_metrics(o['gen_imgs'], target)
            with torch.no_grad():
                # one untimed pass over this rank's images: after the seconds of host-side model construction above the GPU
                # has clocked down, and the first ~50 ms of load run at half speed (tools/time_c3.py: 5.1 ms per image in a
                # first pass, 2.3 ms in every later one); a 2,824-image evaluation is steady state
                sharded_eval.evaluate_sharded(unit, min(n_units, 64 * world), rank, world, device=dev)
                barrier()
                t0 = time.perf_counter()
                table = sharded_eval.evaluate_sharded(unit, n_units, rank, world, device=dev)   # one all_gather at the end
                own3 = time.perf_counter() - t0                    # this rank's own time for its shard (incl. the gather)
                barrier()
                e3 = max_over_ranks(time.perf_counter() - t0)
                n_mine3 = len(range(rank, n_units, world))
                rate3 = [n / t for n, t in zip(all_ranks(float(n_mine3)), all_ranks(own3))]
            assert tuple(table.shape) == (n_units, 8) and torch.isfinite(table).all()
            result["c3"] = {"images": n_units, "images_per_rank": c3_per_rank, "images_per_s": n_units / e3, "wall_s": e3,
                            "per_rank_images_per_s_min": min(rate3), "per_rank_images_per_s_max": max(rate3),
                            "ms_per_image_per_gpu": 1e3 * e3 / c3_per_rank, "mean_psnr": float(table[:, 5].mean()),
                            "mean_ssim": float(table[:, 6].mean()),
                            "note": "BASELINE configs[2]: per image pass #1 render + texture head on (64,64,24,301) local features + "
                                    "pass #2 render + decoder 64^2->1024^2 (cm=2) + 8 metric scalars (ArcFace / LPIPS terms need "
                                    "pretrained nets: reported as 0); images i -> rank i mod W, one all_gather of the (n,8) rows.  "
                                    "Pass #2 reads pass #1's layer-7 record (same styles and poses; see inversion_fwd_note); "
                                    "no_reuse = the same evaluation with pass #2 as a full render launch (E3DGE_REUSE_BACKBONE=0)"}
            if os.environ.get("E3DGE_REUSE_BACKBONE", "1") != "0":
                os.environ["E3DGE_REUSE_BACKBONE"] = "0"
                try:
                    with torch.no_grad():
                        barrier()
                        t0 = time.perf_counter()
                        table0 = sharded_eval.evaluate_sharded(unit, n_units, rank, world, device=dev)
                        barrier()
                        e30 = max_over_ranks(time.perf_counter() - t0)
                    result["c3"]["no_reuse"] = {"images_per_s": n_units / e30, "wall_s": e30, "ms_per_image_per_gpu": 1e3 * e30 / c3_per_rank,
                                                "tables_identical": bool(torch.equal(table0, table))}
                finally:
                    os.environ.pop("E3DGE_REUSE_BACKBONE", None)
        except Exception as exc:  # the headline metric must still be printed
            result["c3"] = {"failed": f"{type(exc).__name__}: {exc}"}

    # ---------------------------------------------------------------- second-pass local features from feature maps
    if rank == 0 and not args.no_inversion:
        try:
            from e3dge_amd.local_query import Fuse_sft_MLP, local_features_from_maps

            def ev_ms2(fn, n=5, rounds=3):
                fn()
                ts = []
                for _ in range(rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / n)
                return statistics.median(ts)
            with torch.no_grad():
                g0 = torch.Generator().manual_seed(5)
                fuse = Fuse_sft_MLP()
                for prm in fuse.parameters():
                    prm.copy_(torch.randn(prm.shape, generator=g0) * (0.1 if prm.ndim == 1 else 1.0 / prm.shape[1] ** 0.5))
                fuse = fuse.to(dev).eval()
                Hh, Ss = 64, 24
                maps = {k: torch.randn(1, 256, 128, 128, generator=g0).to(dev) for k in ("ref", "que")}
                cal = torch.tensor([[[60.0, 0, 0, 64], [0, 60.0, 0, 64], [0, 0, 1, 0]]], device=dev)
                pts = (torch.rand(1, Hh, Hh, Ss, 3, generator=g0) - 0.5).to(dev)
                ldb = {'feature_maps': maps, 'ref_calibs': cal, 'que_calibs': cal, 'points': pts,
                       'xyz': (torch.rand(1, 3, Hh, Hh, generator=g0) - 0.5).to(dev), 'fuse_sft_block': fuse}
                enc_in = torch.randn(1, Hh * Hh * Ss, 513, generator=g0).to(dev)
                spin(args.prewarm_ms / 4)
                t_nat = ev_ms2(lambda: fuse.fuse(enc_in, enc_in[..., 257:]))
                t_all = ev_ms2(lambda: local_features_from_maps(ldb))
                os.environ["E3DGE_FUSE"] = "torch"
                try:
                    t_lib = ev_ms2(lambda: fuse.fuse(enc_in, enc_in[..., 257:]))
                finally:
                    os.environ.pop("E3DGE_FUSE", None)
            n_p = Hh * Hh * Ss
            result["local_features"] = {
                "points": n_p, "from_maps_ms": t_all, "fuse_sft_mlp_ms": t_nat, "fuse_sft_mlp_torch_ms": t_lib,
                "fuse_algorithmic_tflops": 2.0 * 9 * 65536 * n_p / t_nat / 1e9,
                "fuse_hbm_GBps": (513 + 9 * 256 + 8 * 256 + 5 * 256) * 4.0 * n_p / t_nat / 1e6,
                "note": "que_render_given_ref's per-point features from two (1,256,128,128) maps: 3 gathers + Fuse_sft_MLP (nine "
                        "e3dge_ws_linear launches: 590 k MAC per point; bytes = what the nine launches read and write) + positional "
                        "encoding -> (1,64,64,24,301); the torch figure is the same module through library GEMMs (E3DGE_FUSE=torch)"}
        except Exception as exc:
            result["local_features"] = {"failed": f"{type(exc).__name__}: {exc}"}

    # ---------------------------------------------------------------- full inversion forward, one image
    if rank == 0 and not args.no_inversion:
        try:
            with torch.no_grad():
                w1, d1 = syn.synthetic_inputs(1, seed=1, device=dev)
                spin(args.prewarm_ms / 2)
                for _ in range(3):
                    inversion(w1, d1)
                torch.cuda.synchronize()
                ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t1 = time.perf_counter()
                n_inv = 10
                ev_a.record()
                for _ in range(n_inv):
                    o = inversion(w1, d1)
                ev_b.record()
                torch.cuda.synchronize()
                result["inversion_fwd_ms"] = 1e3 * (time.perf_counter() - t1) / n_inv
                result["inversion_fwd_events_ms"] = ev_a.elapsed_time(ev_b) / n_inv
                if os.environ.get("E3DGE_REUSE_BACKBONE", "1") != "0":
                    # the same forward with pass #2 as a full render launch (what rounds 1-2 measured)
                    os.environ["E3DGE_REUSE_BACKBONE"] = "0"
                    try:
                        for _ in range(3):
                            inversion(w1, d1)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(n_inv):
                            inversion(w1, d1)
                        torch.cuda.synchronize()
                        result["inversion_fwd_no_reuse_ms"] = 1e3 * (time.perf_counter() - t1) / n_inv
                    finally:
                        os.environ.pop("E3DGE_REUSE_BACKBONE", None)
                result["inversion_fwd_note"] = ("pass#1 render + texture head on (64,64,24,301) local features + pass#2 render with the "
                                                "resulting texture FiLM + decoder 64^2->1024^2; encoder and the local branch's image "
                                                "filters excluded (out of scope).  Pass #2 sees the same styles and poses as pass #1, and the "
                                                "texture FiLM enters behind the sdf head: it reads pass #1's layer-7 record instead of "
                                                "recomputing layers 0..7 + sdf head + transmittance scan (bit-identical outputs, "
                                                "tests/test_gpu_texhead.py); inversion_fwd_no_reuse_ms is the same forward with pass #2 as a "
                                                "full launch (E3DGE_REUSE_BACKBONE=0)")
                assert tuple(o['gen_imgs'].shape) == (1, 3, 1024, 1024)
                ref_img = o['gen_imgs'].clone()
                result["inversion"] = {"kernels": inversion_kernel_table(gl, w1, d1, o), "note": (
                    "HIP-event time of every launch group of one inversion forward (median of 10), against the roofline that bounds "
                    "it: mfma = algorithmic FLOPs / (dense f16 MFMA peak / 3: split-f16 needs three products), hbm = algorithmic "
                    "bytes / 8 TB/s; `bound` is the larger of the two minimum times, frac = that minimum time / measured time")}
            # the decoder and the SFT block with a graph wanted (features / latent / inputs require grad, parameters frozen: the shape of
            # a train_ae.py forward): autograd nodes with the native forward vs the library path
            try:
                def ev_fb(fn, n=5):
                    for _ in range(2):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(); e0.record()
                    for _ in range(n):
                        fn()
                    e1.record(); torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / n
                dec_ = gl.decoder
                fm_ = o['features'].detach()
                ag = {}
                # stage-1 shape (trainer.py:1017-1031, scripts/train/ffhq/stage1.sh --E_d_grad_false --disable_decoder_fpn): the feature map
                # requires grad, the decoder latent and the generator do not; pixel loss on pool_256(gen_imgs).  "packed" = the default
                # (packed forward + e3dge_dec2_backward), "library" = weight modulation + MIOpen for both directions (round 4's default)
                pool_ = torch.nn.AdaptiveAvgPool2d((256, 256))
                for be, env in (("packed", "auto"), ("library", "library")):
                    os.environ["E3DGE_DECODER_AUTOGRAD"] = env
                    try:
                        def fb():
                            f_ = fm_.clone().requires_grad_(True)
                            pool_(dec_(f_, [d1], input_is_latent=True, randomize_noise=False)[0]).square().mean().backward()
                        ag["decoder_" + be + "_fwd_bwd_ms"] = ev_fb(fb)
                        if be == "packed":      # roofline of the decoder's forward + data-gradient backward (nine 3x3 layers each way, 125.6 GFLOP each)
                            t_ = ag["decoder_packed_fwd_bwd_ms"]
                            ag["decoder_packed_fwd_bwd_roofline"] = {
                                "bound": "mfma", "achieved": 2 * 125.6e9 / (t_ * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3, "unit": "TFLOP/s",
                                "frac": 2 * 125.6e9 / (PEAK_F16_MFMA_TFLOPS / 3 * 1e12) / (t_ * 1e-3), "flop": 2 * 125.6e9,
                                "note": "e3dge_dec2_forward + e3dge_dec2_backward + the caller's pool_256 and loss; event time of forward + backward"}
                    finally:
                        os.environ.pop("E3DGE_DECODER_AUTOGRAD", None)
                try:        # d latent wanted too (not the stage-1 shape): no native backward for it, the library path runs
                    def fb_l():
                        f_, l_ = fm_.clone().requires_grad_(True), d1.detach().clone().requires_grad_(True)
                        pool_(dec_(f_, [l_], input_is_latent=True, randomize_noise=False)[0]).square().mean().backward()
                    ag["decoder_with_d_latent_fwd_bwd_ms"] = ev_fb(fb_l)
                except Exception as exc:                                  # noqa: BLE001
                    ag["decoder_with_d_latent_fwd_bwd_ms"] = f"failed: {type(exc).__name__}"
                try:        # every launch of one packed backward (HIP events inside the native call)
                    nz_ = [getattr(dec_.noises, f"noise_{i}") for i in range(dec_.num_layers)]
                    gy_ = torch.randn(1, 3, 1024, 1024, device=dev) / 1024
                    kb_ = []
                    with torch.no_grad():
                        for _ in range(3):
                            dec_._forward_packed(fm_, d1, nz_, save=True)
                            dec_._backward_packed(fm_, gy_, kernel_ms=kb_)
                    # (the 21 per-launch figures are in profiles/r5_decoder_autograd.json = tools/time_decoder_autograd.py; the line carries the sum
                    # and the three largest, to stay within the tail the driver records)
                    top3 = sorted(zip(kb_, dec_.dec2_bwd_launch_names()), reverse=True)[:3]
                    ag["decoder_backward_sum_of_launches_ms"] = round(sum(kb_), 4)
                    ag["decoder_backward_longest_launches_ms"] = {n: round(v, 4) for v, n in top3}
                except Exception as exc:                                  # noqa: BLE001
                    ag["decoder_backward_sum_of_launches_ms"] = f"failed: {type(exc).__name__}: {exc}"[:160]
                from e3dge_amd.local_query import Fuse_sft_MLP
                fu_ = Fuse_sft_MLP().to(dev)
                fu_.requires_grad_(False)
                with torch.no_grad():
                    for prm in fu_.parameters():
                        prm.copy_(torch.randn_like(prm) * (0.1 if prm.ndim == 1 else 1.0 / prm.shape[1] ** 0.5))
                xin_ = torch.randn(1, RES * RES * N_SAMPLES, 513, device=dev)
                for be in ("hip", "torch"):
                    os.environ["E3DGE_FUSE_AUTOGRAD"] = be
                    try:
                        def fb2():
                            x_ = xin_.clone().requires_grad_(True)
                            fu_.fuse(x_, x_[..., 257:]).square().mean().backward()
                        ag["fuse_sft_" + be + "_fwd_bwd_ms"] = ev_fb(fb2)
                        fu_.requires_grad_(True)          # stage 2 trains this module: + the thirteen parameter gradients (e3dge_wgrad vs matmul)
                        ag["fuse_sft_" + be + "_trainable_fwd_bwd_ms"] = ev_fb(fb2)
                    finally:
                        fu_.requires_grad_(False)
                        os.environ.pop("E3DGE_FUSE_AUTOGRAD", None)
                try:        # the texture head (ResnetBlockFC 301 -> 512 on 98,304 points) under autograd: native forward, e3dge_tex_modulations_bwd (round 5) vs library GEMMs
                    head_ = gl.renderer.network.netLocal.local_feat_to_tex_modulations_linear
                    f_h = feats.detach().clone()

                    def fb3():
                        x_ = f_h.clone().requires_grad_(True)
                        al_, be_ = head_.tex_modulations(x_)
                        (al_.square().mean() + be_.square().mean()).backward()
                    ag["tex_head_fwd_bwd_ms"] = ev_fb(fb3)
                    head_.requires_grad_(True)
                    ag["tex_head_trainable_fwd_bwd_ms"] = ev_fb(fb3)
                    os.environ["E3DGE_TEXHEAD_BWD"] = "library"
                    try:
                        ag["tex_head_library_trainable_fwd_bwd_ms"] = ev_fb(fb3)
                        head_.requires_grad_(False)
                        ag["tex_head_library_fwd_bwd_ms"] = ev_fb(fb3)
                    finally:
                        head_.requires_grad_(False)
                        os.environ.pop("E3DGE_TEXHEAD_BWD", None)
                    with torch.no_grad():
                        ag["tex_head_fwd_ms"] = ev_fb(lambda: head_.tex_modulations(f_h))
                except Exception as exc:                                  # noqa: BLE001
                    ag["tex_head_fwd_bwd_ms"] = f"failed: {type(exc).__name__}: {exc}"[:160]
                # ---- the stage-2 training step end to end (round 6): que_render_given_ref under autograd, both trainable modules ----
                # (e3dge_full_runner.py:185-317: first render of the query view, gathers on the two feature maps, Fuse_sft_MLP, PE, texture
                #  head, second render with the per-point FiLM, decoder, pixel loss on pool_256 + thumbnail loss; backward into the feature
                #  maps, Fuse_sft_MLP, the texture head and the renderer latent.  Pinned against the reference's own autograd at 16x16x24 by
                #  tests/test_gpu_stage2.py.)
                try:
                    fu2 = Fuse_sft_MLP().to(dev)
                    with torch.no_grad():
                        for prm in fu2.parameters():
                            prm.copy_(torch.randn_like(prm) * (0.1 if prm.ndim == 1 else 1.0 / prm.shape[1] ** 0.5))
                    fu2.requires_grad_(True)
                    head_ = gl.renderer.network.netLocal.local_feat_to_tex_modulations_linear
                    g2_ = torch.Generator(device=dev).manual_seed(11)
                    maps2 = {k: torch.randn(1, 256, 128, 128, device=dev, generator=g2_).requires_grad_(True) for k in ("ref", "que")}
                    cq_ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev), return_calibs=True)['calibs']
                    cr_ = generate_camera_params(RES, dev, locations=torch.tensor([[-0.2, 0.05]], device=dev), return_calibs=True)['calibs']
                    pool2 = torch.nn.AdaptiveAvgPool2d((256, 256))

                    def stage2_step():
                        with torch.no_grad():
                            o1 = gl.renderer(p1, f1, n1, fa1, styles=w1)                       # the query view's first render: points, xyz
                        s_ = w1.clone().requires_grad_(True)
                        for t_ in list(maps2.values()) + list(fu2.parameters()) + list(head_.parameters()):
                            t_.grad = None
                        o2 = gl([s_, d1], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False,
                                local_data_batch=dict(feature_maps=maps2, ref_calibs=cr_, que_calibs=cq_, points=o1['points'], xyz=o1['xyz'],
                                                      fuse_sft_block=fu2))
                        ((pool2(o2['gen_imgs']) ** 2).mean() + (o2['gen_thumb_imgs'] ** 2).mean()).backward()
                        return s_.grad
                    head_.requires_grad_(True)
                    try:
                        gs2 = stage2_step()
                        assert torch.isfinite(gs2).all() and all(m_.grad is not None and torch.isfinite(m_.grad).all() for m_ in maps2.values())
                        assert all(p_.grad is not None for p_ in fu2.parameters()) and all(p_.grad is not None for p_ in head_.parameters())
                        ts2 = sorted(ev_fb(stage2_step, n=4) for _ in range(5))
                    finally:
                        head_.requires_grad_(False)
                    n_p2 = RES * RES * N_SAMPLES
                    fl2 = dict(render_first=FLOP_PER_RAY * RES * RES, render_second_saving=FLOP_PER_RAY * RES * RES,
                               render_backward=8 * 131072.0 * n_p2, fuse_fwd_bwd_wgrad=3 * 2.0 * 9 * 65536 * n_p2,
                               head_fwd_bwd_wgrad=3 * 2.0 * 398825 * n_p2, decoder_fwd_bwd=2 * 125.6e9)
                    fl2_tot = sum(fl2.values())
                    result["train_step_stage2_ms"] = ts2[len(ts2) // 2]
                    result["train_step_stage2"] = {
                        "step_ms": ts2[len(ts2) // 2], "step_ms_min": ts2[0], "estimator": "median of 5 blocks of 4 steps, HIP events",
                        "roofline": {"bound": "mfma", "achieved": fl2_tot / (ts2[len(ts2) // 2] * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3,
                                     "unit": "TFLOP/s", "frac": fl2_tot / (PEAK_F16_MFMA_TFLOPS / 3 * 1e12) / (ts2[len(ts2) // 2] * 1e-3),
                                     "flop_per_step": fl2_tot, "flop_by_part_G": {k_: round(v_ / 1e9, 1) for k_, v_ in fl2.items()}},
                        "note": "one stage-2 sample, 64x64x24 points, two (1,256,128,128) feature maps, Fuse_sft_MLP and the texture head trainable, "
                                "generator frozen: first render + 3 gathers + Fuse_sft_MLP + PE + texture head + texture-FiLM render (saving) + decoder "
                                "1024^2 + loss on pool_256(image) and the thumbnail; backward through all of it (e3dge_dec2_backward, "
                                "e3dge_siren_render_bwd in the 8-wave TEX form, e3dge_tex_modulations_bwd + e3dge_wgrad, the Fuse_sft_MLP node, "
                                "e3dge_local_query_bwd).  The hourglass image filters are outside the path (SURVEY 2)."}
                    del fu2, maps2
                except Exception as exc:                                  # noqa: BLE001
                    result["train_step_stage2_ms"] = None
                    result["train_step_stage2"] = {"failed": f"{type(exc).__name__}: {exc}"[:240]}
                ag["note"] = ("forward + backward with a graph: decoder in the stage-1 shape (features require grad, latent + parameters frozen, loss on "
                              "pool_256(image)): packed = e3dge_dec2_forward + e3dge_dec2_backward, library = weight modulation + MIOpen; Fuse_sft_MLP: "
                              "native-forward autograd node vs torch modules")
                result["autograd"] = ag
            except Exception as exc:
                result["autograd"] = {"failed": f"{type(exc).__name__}: {exc}"}
            # the same ~60 launches replayed as one HIP graph (cvpr23-e3dge_amd/graphs.py)
            try:
                from e3dge_amd.graphs import GraphedCall
                gi = GraphedCall(lambda a, b: inversion(a, b)['gen_imgs'], w1, d1)
                img_g = gi(w1, d1)
                torch.cuda.synchronize()
                spin(args.prewarm_ms / 2)        # the same clock state as the eager figure above (the autograd leg in between is host-bound: the GPU clocks down)
                # the first replays of a freshly instantiated graph are slower than the rest (tools/graph_vs_eager.py: blocks of ten replays
                # right after the capture); rounds 3-4 timed exactly those -- which is where "a replay slower than 21 eager launches" came from
                e_c0, e_c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e_c0.record()
                for _ in range(n_inv):
                    gi(w1, d1)
                e_c1.record()
                torch.cuda.synchronize()
                result["inversion_fwd_graph_first10_ms"] = e_c0.elapsed_time(e_c1) / n_inv
                t1 = time.perf_counter()
                ev_a.record()
                for _ in range(n_inv):
                    img_g = gi(w1, d1)
                ev_b.record()
                torch.cuda.synchronize()
                result["inversion_fwd_graph_ms"] = 1e3 * (time.perf_counter() - t1) / n_inv
                result["inversion_fwd_graph_events_ms"] = ev_a.elapsed_time(ev_b) / n_inv
                # steady state: 50 back-to-back replays between two events (the 10-iteration wall clock above carries the fixed cost of the
                # first replay after a synchronize -- the runtime re-arms the graph's ~25 packets before anything runs -- and the two
                # input copies per call; tools/graph_vs_eager.py)
                ev_a.record()
                for _ in range(50):
                    gi.graph.replay()
                ev_b.record()
                torch.cuda.synchronize()
                result["inversion_fwd_graph_steady_ms"] = ev_a.elapsed_time(ev_b) / 50
                result["inversion_fwd_graph_max_abs_diff_vs_eager"] = float((img_g - ref_img).abs().max())
                del gi
            except Exception as exc:
                result["inversion_fwd_graph_ms"] = None
                result["inversion_fwd_graph_note"] = f"capture failed: {type(exc).__name__}: {exc}"
        except Exception as exc:
            result["inversion_fwd_ms"] = None
            result["inversion_fwd_note"] = f"failed: {type(exc).__name__}: {exc}"
    dec_keep = gl.decoder if gl is not None else None          # (the full stage-1 step below runs the decoder again)
    del gl

    # ---------------------------------------------------------------- C4: the 120-pose sweep at 128x128 rays x 48 samples
    if not args.no_c4:
        try:
            import math
            r4 = VolumeFeatureRenderer(syn.rendering_opt(N_samples=48), out_im_res=128, mode='test')
            r4.load_state_dict(renderer.state_dict())
            r4 = r4.to(dev)
            w4, _ = syn.synthetic_inputs(1, seed=1, device=dev)
            n_pose = 120                                                   # trainer.py:2349-2388: azim_k = 0.45 cos(pi k / 119)
            traj = torch.tensor([[0.45 * math.cos(math.pi * k / (n_pose - 1)), 0.0] for k in range(n_pose)], device=dev)
            mine4 = list(sharded_eval.shard_indices(n_pose, rank, world))     # pose k -> rank k mod W
            p4, f4, n4, fa4, _ = generate_camera_params(128, dev, locations=traj[mine4])
            n_mine = len(mine4)

            def sweep(bsz):
                wb = w4.expand(bsz, -1, -1).contiguous()
                film = r4.siren.film_params(wb)                           # one latent for the whole sweep
                o = None
                for k in range(0, n_mine, bsz):
                    o = r4.render_with_film(film[:min(bsz, n_mine - k)], f4[k:k + bsz], p4[k:k + bsz], n4[k:k + bsz], fa4[k:k + bsz])
                return o
            c4 = {}
            spin(args.prewarm_ms / 2)
            with torch.no_grad():
                # (a batch-of-8 leg used to run beside this: 2.340 vs 2.293 ms per pose in round 3 -- one 128x128x48 pose is 6,144
                # sub-tiles, 24 per CU, so batching poses buys nothing; dropped, SURVEY's "B = 8" variant is covered by the parity tests)
                for label, bsz in (("sequential", 1),):
                    sweep(bsz)
                    barrier()
                    t0 = time.perf_counter()
                    sweep(bsz)
                    barrier()
                    ms = 1e3 * max_over_ranks(time.perf_counter() - t0)      # wall time of the whole 120-pose sweep over all ranks
                    c4[label] = {"sweep_ms": ms, "ms_per_pose": ms / n_pose, "rays_per_s": n_pose * 128 * 128 / ms * 1e3,
                                 "algorithmic_tflops": 2 * MAC_PER_POINT * 48 * 128 * 128 * n_pose / (ms * 1e-3) / 1e12}
            c4["poses_per_rank"] = [int(x) for x in all_ranks(float(n_mine))]
            c4["note"] = ("BASELINE configs[3]: 120 camera poses of one latent, 128x128 rays x 48 samples (786,432 points per pose), "
                          "pose k -> rank k mod W, wall time between barriers")
            result["c4"] = c4
            del r4
        except Exception as exc:
            result["c4"] = {"failed": f"{type(exc).__name__}: {exc}"}

    # ---------------------------------------------------------------- surface extraction: 128^3 SDF volume + align_volume
    if rank == 0 and not args.no_surface:
        try:
            from e3dge_amd import mesh_utils
            rs_ = VolumeFeatureRenderer(syn.rendering_opt(N_samples=128), out_im_res=128, mode='test')   # train_setup.py:112-126
            rs_.load_state_dict(renderer.state_dict())
            rs_ = rs_.to(dev)
            ws_, _ = syn.synthetic_inputs(1, seed=1, device=dev)
            ps_, fs_, ns_, fas_, _ = generate_camera_params(128, dev, locations=torch.zeros(1, 2, device=dev))
            with torch.no_grad():
                def extract():
                    o = rs_(ps_, fs_, ns_, fas_, styles=ws_)
                    return mesh_utils.align_volume(o['sdf'])
                spin(args.prewarm_ms / 2)
                for _ in range(2):
                    extract()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                torch.cuda.synchronize()
                ev[0].record()
                for _ in range(5):
                    o = rs_(ps_, fs_, ns_, fas_, styles=ws_)
                ev[1].record()
                for _ in range(5):
                    mesh_utils.align_volume(o['sdf'])
                ev[2].record()
                torch.cuda.synchronize()
                ms_r, ms_a = ev[0].elapsed_time(ev[1]) / 5, ev[1].elapsed_time(ev[2]) / 5
            result["surface"] = {"render_ms": ms_r, "align_volume_ms": ms_a, "points": 128 ** 3,
                                 "points_per_s": 128 ** 3 / ms_r * 1e3,
                                 "algorithmic_tflops": 2 * MAC_PER_POINT * 128 ** 3 / (ms_r * 1e-3) / 1e12,
                                 "align_volume_GBps": 8 * 128 ** 3 / (ms_a * 1e-3) / 1e9,
                                 "note": "surf_extraction generator (train_setup.py:112-126): 128x128 rays x 128 samples, then "
                                         "align_volume (mesh_utils.py:17-44); marching cubes (CPU, third-party) not included"}
            del rs_
        except Exception as exc:
            result["surface"] = {"failed": f"{type(exc).__name__}: {exc}"}

    # ---------------------------------------------------------------- the two stream ops at the decoder's top-level size (HBM fractions)
    if rank == 0 and not args.no_inversion:
        try:
            from e3dge_amd import op as e3op
            from e3dge_amd.stylesdf_model import make_kernel
            k4_ = (make_kernel([1, 3, 3, 1]) * 4).to(dev)

            def ev_op(fn, n=30):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(n):
                    fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e-3
            so = {}
            with torch.no_grad():
                xb = torch.randn(1, 32, 1025, 1025, device=dev)
                xa = torch.randn(1, 32, 1024, 1024, device=dev)
                bb = torch.randn(32, device=dev)
                for tag, xb_, xa_, bb_, bpe in (("f32", xb, xa, bb, 4), ("f16", xb.half(), xa.half(), bb.half(), 2)):
                    t_blur = ev_op(lambda: e3op.upfirdn2d(xb_, k4_, pad=(1, 1)))
                    t_act = ev_op(lambda: e3op.fused_leaky_relu(xa_, bb_))
                    so["blur_" + tag] = {"us": 1e6 * t_blur, "hbm_frac": bpe * (xb_.numel() + 32 * 1024 * 1024) / t_blur / (PEAK_HBM_GBPS * 1e9)}
                    so["bias_act_" + tag] = {"us": 1e6 * t_act, "hbm_frac": 2 * bpe * xa_.numel() / t_act / (PEAK_HBM_GBPS * 1e9)}
                del xb, xa
            so["note"] = "Blur (upfirdn2d 4x4) on (1,32,1025,1025), fused_leaky_relu on (1,32,1024,1024): (in + out) bytes / event time / 8 TB/s"
            result["stream_ops"] = so
        except Exception as exc:                                          # noqa: BLE001
            result["stream_ops"] = {"failed": f"{type(exc).__name__}: {exc}"[:160]}

    # ---------------------------------------------------------------- C5: stage-1 training step of the renderer
    if not args.no_train_step:
        try:
            S5 = 18                                                    # scripts/train/ffhq/stage1.sh
            r5 = VolumeFeatureRenderer(syn.rendering_opt(N_samples=S5), out_im_res=RES, mode='test')
            r5.load_state_dict(renderer.state_dict())
            r5 = r5.to(dev)
            r5.requires_grad_(False)                                   # frozen generator, gradient to the styles only
            w5, _ = syn.synthetic_inputs(1, seed=1 + rank, device=dev)
            p5, f5, n5, fa5, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))

            def train_step():
                s_ = w5.clone().requires_grad_(True)
                o = r5(p5, f5, n5, fa5, styles=s_, return_eikonal=True, return_surface_eikonal=True)
                loss = ((o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean()
                        + (o['surface_eikonal_term'] ** 2).mean())
                loss.backward()
                return s_.grad

            def wall_ms(fn, n):
                barrier()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                barrier()
                return 1e3 * max_over_ranks(time.perf_counter() - t0) / n

            def block_stats(fn, steps, blocks=5):
                """The same estimator for every training leg (round 6): `blocks` blocks of `steps` steps, wall clock between barriers, max
                over ranks per block; the MEDIAN block is the figure, min / max go on the line beside it."""
                v = sorted(wall_ms(fn, steps) for _ in range(blocks))
                return {"median": statistics.median(v), "min": v[0], "max": v[-1], "blocks": blocks, "steps_per_block": steps}
            # (round 5 read this step at 3.27 instead of 2.53 ms on two boxes.  It is the clock: the GPU clocks down while the host is busy --
            # a process start, the host-bound autograd legs above -- and a handful of 2.5-ms warm-up steps does not bring it back: every
            # process but the first on a fresh box reads tools/c5_step.py at 3.4-4.1 ms after three warm-up steps and at 2.51 after 0.4 s
            # of them.  Hence the spin below; the empty_cache() was the first suspect and stays.)
            torch.cuda.empty_cache()
            spin(args.prewarm_ms / 2)
            for _ in range(5):
                gr = train_step()
            n_tr = 4
            st1 = block_stats(train_step, n_tr)
            ms = st1["median"]
            assert torch.isfinite(gr).all()
            # the fp32 fallback of the same step (what a checkpoint with |w| >= 256 takes): two of its kernels still spill (20 / 60 B
            # of scratch, tools/scratch_report.sh) -- its cost goes on the line instead of being implied
            try:
                m0, b0 = r5.siren.mfma_mode, r5.siren.bwd_mode
                r5.siren.mfma_mode = r5.siren.bwd_mode = "f32"
                for _ in range(2):
                    train_step()
                ms_f32 = block_stats(train_step, 2, 3)["median"]
                r5.siren.mfma_mode, r5.siren.bwd_mode = m0, b0
                if isinstance(result.get("modes"), dict) and "f32" in result["modes"]:
                    result["modes"]["f32"]["train_step_ms"] = ms_f32
                result["train_step_f32_fallback_ms"] = ms_f32
            except Exception as exc:
                result["train_step_f32_fallback_ms"] = f"failed: {type(exc).__name__}: {exc}"[:120]
            # scripts/train/ffhq/stage1.sh trains with batch_size=4 PER GPU; BASELINE configs[4] (bs=8 on 8 GPUs) is one sample per GPU, and at
            # 64x64x18 one sample is 576 workgroup tiles of 128 points = 2.25 rounds of the 256 CUs, i.e. 3 rounds with the last one a
            # quarter full.  Four samples are 9 full rounds: the same kernels, per sample (supplement, not the C5 number)
            try:
                w5b, _ = syn.synthetic_inputs(4, seed=11 + rank, device=dev)
                p5b, f5b, n5b, fa5b, _ = generate_camera_params(RES, dev, locations=torch.zeros(4, 2, device=dev))

                def train_step_b4():
                    s_ = w5b.clone().requires_grad_(True)
                    o = r5(p5b, f5b, n5b, fa5b, styles=s_, return_eikonal=True, return_surface_eikonal=True)
                    loss = ((o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean()
                            + (o['surface_eikonal_term'] ** 2).mean())
                    loss.backward()
                    return s_.grad
                for _ in range(3):
                    g4 = train_step_b4()
                st4 = block_stats(train_step_b4, 3)
                result["train_step_batch4_ms_per_sample"] = st4["median"] / 4
                result["train_step_batch4_ms_per_sample_min"] = st4["min"] / 4
                assert torch.isfinite(g4).all()
            except Exception as exc:                                      # noqa: BLE001
                result["train_step_batch4_ms_per_sample"] = f"failed: {type(exc).__name__}: {exc}"[:160]
            # the step train_ae.py actually runs for one sample (scripts/train/ffhq/stage1.sh: --full_pipeline): renderer forward with the
            # eikonal terms -> decoder 64^2 -> 1024^2 -> pixel loss on pool_256(gen_imgs) (trainer.py:1017-1031) + the renderer losses,
            # backward through the decoder (d features only: latent and generator frozen) into the renderer, down to the styles
            try:
                # through the generator's own entry point, as trainer.py:881-897 calls it (renderer -> decoder in one forward)
                g5 = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S5), full_pipeline=True)
                syn.load_synthetic(g5)
                g5 = g5.to(dev).eval()
                g5.requires_grad_(False)
                _, d5 = syn.synthetic_inputs(1, seed=1 + rank, device=dev)
                pool5 = torch.nn.AdaptiveAvgPool2d((256, 256))
                full = {}

                def train_step_full(latent_grad=False):
                    s_ = w5.clone().requires_grad_(True)
                    dl_ = d5.clone().requires_grad_(True) if latent_grad else d5
                    o = g5([s_, dl_], p5, f5, n5, fa5, input_is_latent=True, randomize_noise=False, return_eikonal=True,
                           return_surface_eikonal=True)
                    loss = ((pool5(o['gen_imgs']) ** 2).mean() + (o['gen_thumb_imgs'] ** 2).mean()
                            + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean() + (o['surface_eikonal_term'] ** 2).mean())
                    loss.backward()
                    return s_.grad
                for be, env, ov in (("packed", "auto", "1"), ("library", "library", "1"), ("packed_no_overlap", "auto", "0")):
                    os.environ["E3DGE_DECODER_AUTOGRAD"] = env
                    os.environ["E3DGE_OVERLAP_DECODER"] = ov
                    try:
                        for _ in range(3):
                            gf = train_step_full()
                        full_st = block_stats(train_step_full, n_tr)
                        full[be], full[be + "_min"] = full_st["median"], full_st["min"]
                        assert torch.isfinite(gf).all()
                    finally:
                        os.environ.pop("E3DGE_DECODER_AUTOGRAD", None)
                        os.environ.pop("E3DGE_OVERLAP_DECODER", None)
                result["train_step_full_no_overlap_ms"] = full["packed_no_overlap"]
                # both latents trainable (what the encoder's two heads receive, trainer.py:881-897): d latent from e3dge_dec2_backward too
                for be, env in (("packed", "auto"), ("library", "library")):
                    os.environ["E3DGE_DECODER_AUTOGRAD"] = env
                    try:
                        for _ in range(3):
                            gf = train_step_full(True)
                        full[be + "_both"] = block_stats(lambda: train_step_full(True), n_tr)["median"]
                        assert torch.isfinite(gf).all()
                    finally:
                        os.environ.pop("E3DGE_DECODER_AUTOGRAD", None)
                result["train_step_full_both_latents_ms"] = full["packed_both"]
                result["train_step_full_both_latents_library_ms"] = full["library_both"]
                # and at stage1.sh's batch_size = 4 per GPU (both latents trainable): per sample
                try:
                    _, d5b = syn.synthetic_inputs(4, seed=21 + rank, device=dev)

                    def train_step_full_b4():
                        s_, dl_ = w5b.clone().requires_grad_(True), d5b.clone().requires_grad_(True)
                        o = g5([s_, dl_], p5b, f5b, n5b, fa5b, input_is_latent=True, randomize_noise=False, return_eikonal=True,
                               return_surface_eikonal=True)
                        loss = ((pool5(o['gen_imgs']) ** 2).mean() + (o['gen_thumb_imgs'] ** 2).mean()
                                + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean() + (o['surface_eikonal_term'] ** 2).mean())
                        loss.backward()
                        return s_.grad
                    for _ in range(4):
                        gf = train_step_full_b4()
                    # (13 GB of new blocks at this batch: one bench run read 28.6 ms per step over five steps right after two warm-ups, the
                    # stand-alone tools/full_step_b4.py 13.5 ms from its second step on -- the allocator was still growing.  Five blocks of
                    # three steps like every other leg: median, the fastest block beside it.)
                    stf4 = block_stats(train_step_full_b4, 3)
                    result["train_step_full_batch4_ms_per_sample"] = stf4["median"] / 4
                    result["train_step_full_batch4_ms_per_sample_min"] = stf4["min"] / 4
                    assert torch.isfinite(gf).all()
                except Exception as exc:                                      # noqa: BLE001
                    result["train_step_full_batch4_ms_per_sample"] = f"failed: {type(exc).__name__}: {exc}"[:160]
                del g5
                result["train_step_full_ms"] = full["packed"]
                result["train_step_full_ms_min"] = full["packed_min"]
                result["train_step_full_library_decoder_ms"] = full["library"]
                # roofline of the whole sample (round 6): the renderer's 30 GEMM chains + the decoder's nine 3x3 layers forward and
                # data-gradient backward (125.6 GFLOP each at 1024^2, channel multiplier 2) over the measured step, on f16 MFMA / 3.
                # The caller's pool_256 (~0.16 ms of adaptive_avg_pool forward + backward, profiles/r5_full_step_timeline.txt) is inside
                # the measured time and outside the flop count.
                fl_full = 30 * 131072.0 * (RES * RES * S5 + RES * RES) + 2 * 125.6e9
                result["train_step_full"] = {"step_ms": full["packed"], "roofline": {
                    "bound": "mfma", "achieved": fl_full / (full["packed"] * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3, "unit": "TFLOP/s",
                    "frac": fl_full / (PEAK_F16_MFMA_TFLOPS / 3 * 1e12) / (full["packed"] * 1e-3), "flop_per_step": fl_full,
                    "note": "renderer step (306 GFLOP) + decoder forward + data-gradient backward (2 x 125.6 GFLOP); ~4 % of the measured "
                            "time is the caller's pool_256, counted in the time only"}}
                result["train_step_full_note"] = ("one stage-1 sample as train_ae.py runs it (--full_pipeline): C5 renderer step + decoder 64^2 -> 1024^2 forward "
                                                  "and backward with the pixel loss on pool_256(gen_imgs) (trainer.py:1017-1031); decoder latent and generator "
                                                  "frozen; packed = e3dge_dec2_forward / e3dge_dec2_backward, library = round 4's path; wall clock, max over ranks")
            except Exception as exc:                                      # noqa: BLE001
                result["train_step_full_ms"] = None
                result["train_step_full_note"] = f"failed: {type(exc).__name__}: {exc}"[:200]
            result["train_step_ms"] = ms
            result["train_step_ms_min"] = st1["min"]
            result["train_step_rays_per_sec"] = world * RES * RES / ms * 1e3
            result["train_step_note"] = ("C5 renderer part, 1 image 64x64x18 per GPU: forward saving arguments + eikonal term (sdf chain) + "
                                         "surface normals at the integrated point (kept in the graph), loss = mean(rgb^2) + "
                                         "mean((|eik|-1)^2) + mean(surf_eik^2), backward to the styles incl. the double backward "
                                         f"(tangent + second-order chain); backward mode {r5.siren.bwd_mode}; median of 5 blocks of 4 steps, max over ranks between barriers "
                                         "(the same estimator on every training leg; *_min = the fastest block)")
            # roofline of the step as a whole: 30 GEMM chains of 131,072 FLOP per point (forward 8 + sdf chain 7 + tangent 7
            # + second-order backward 8) over 64*64*18 ray samples + 4,096 surface points, and the saved state -- 9 x 256 pre-sine
            # arguments written once and read three times, r_l and ta_l (8 x 256 each) written and read once = 66 KB per ray sample
            n_samp, n_surf = RES * RES * S5, RES * RES
            flop_step = 30 * 131072.0 * (n_samp + n_surf)
            bytes_step = n_samp * (9 * 256 * 4 * 4 + 2 * 8 * 256 * 4 * 2) + n_surf * (9 * 256 * 4 * 2 + 8 * 256 * 4 * 2)
            t_f, t_b = flop_step / (PEAK_F16_MFMA_TFLOPS / 3 * 1e12), bytes_step / (PEAK_HBM_GBPS * 1e9)
            # The bound of the step is its arithmetic: inputs and outputs are a few MB; the 5.3 GB of saved state
            # (pre-sine arguments, r_l, ta_l) is this implementation's choice and is reported as such, next to the PMC traffic.
            tr_traffic = None
            try:
                with open(TRAFFIC_FILE) as f:
                    tj = json.load(f)
                te = tj.get("train_step")
                if te:
                    tr_traffic = int(2 * te["FETCH_SIZE_KB"] * 1024 + te["WRITE_SIZE_KB"] * 1024)
            except (OSError, ValueError, KeyError):
                pass
            ts = {"ranks": world, "step_ms": ms,
                  "roofline": {"bound": "mfma", "achieved": flop_step / (ms * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3, "unit": "TFLOP/s",
                               "frac": t_f / (ms * 1e-3), "flop_per_step": flop_step, "saved_state_bytes_per_step": bytes_step,
                               "saved_state_hbm_frac": t_b / (ms * 1e-3), "traffic": tr_traffic,
                               "note": "bound = the 30 GEMM chains of the step on f16 MFMA / 3; saved-state bytes are a design cost, not algorithmic; "
                                       "traffic = PMC FETCH x2 + WRITE per step (profiles/traffic_pmc.json)"}}
            b4 = result.get("train_step_batch4_ms_per_sample")
            if isinstance(b4, float):       # the same arithmetic per sample at stage1.sh's four samples per GPU (three full 256-CU rounds per launch instead of 2.25)
                ts["roofline_batch4"] = {"ms_per_sample": b4, "achieved": flop_step / (b4 * 1e-3) / 1e12, "frac": t_f / (b4 * 1e-3),
                                         "estimator": "median of 5 blocks of 3 steps", "ms_per_sample_min": result.get("train_step_batch4_ms_per_sample_min")}
            if dist is not None:
                # stage 1 trains the encoder under DDP -- 1.03 GB of fp32 gradients all-reduced per step
                # (trainer.py:1737-1778, dist_utils.py:108-130).  The encoder is out of scope; its collective is emulated with a
                # bucket of that size on a side stream, issued at the start of the renderer's step as DDP overlaps it with backward.
                bucket = torch.zeros(int(ENCODER_GRAD_BYTES // 4), device=dev)
                side = torch.cuda.Stream(device=dev)

                def ar_only():
                    with torch.cuda.stream(side):
                        dist.all_reduce(bucket)
                    side.synchronize()

                def both():
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        dist.all_reduce(bucket)
                    train_step()
                    torch.cuda.current_stream(dev).wait_stream(side)
                for _ in range(2):
                    ar_only()
                    both()
                t_ar = wall_ms(ar_only, 5)
                t_both = wall_ms(both, 5)
                ts.update(allreduce_bytes=ENCODER_GRAD_BYTES, allreduce_ms=t_ar, step_with_allreduce_ms=t_both,
                          allreduce_busbw_GBps=2 * (world - 1) / world * ENCODER_GRAD_BYTES / (t_ar * 1e-3) / 1e9,
                          overlap_frac=(ms + t_ar - t_both) / min(ms, t_ar))
                del bucket
            result["train_step"] = ts
            del r5
        except Exception as exc:
            result["train_step_ms"] = None
            result["train_step_note"] = f"failed: {type(exc).__name__}: {exc}"

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import renderer_ref
        cores = os.cpu_count() or 1
        c = lambda t: t.detach().cpu()
        a = (c(poses[:1]), c(focal[:1]), c(near[:1]), c(far[:1]), c(wr[:1]))
        trials = {}
        with torch.no_grad():
            # PyTorch's CPU GEMMs stop scaling (and on many-socket hosts collapse) long before all hardware threads
            # are used, so the baseline is taken at 8 threads (the authoring container's count), at 32 and at
            # min(64, cores); the best is reported with the thread count that produced it.
            for nt in sorted({min(8, cores), min(32, cores), min(64, cores)}):
                torch.set_num_threads(nt)
                renderer_ref.render(sd_cpu, *a, res=RES, n_samples=N_SAMPLES)                  # warm-up
                n, t1 = 0, time.perf_counter()
                while True:
                    renderer_ref.render(sd_cpu, *a, res=RES, n_samples=N_SAMPLES)
                    n += 1
                    dt = time.perf_counter() - t1
                    if dt > 5.0 or n >= 12:
                        break
                trials[nt] = (n * RES * RES / dt, n, dt)
        best = max(trials, key=lambda k: trials[k][0])
        result["cpu_baseline"] = {"value": trials[best][0], "unit": "rays/s", "cores": best, "kind": "port",
                                  "host_cores": cores,
                                  "sample": "; ".join(f"{nt} threads: {v[1]} renders of one 64x64x24 image in {v[2]:.1f} s = {v[0]:.0f} rays/s"
                                                      for nt, v in sorted(trials.items())) +
                                  " (oracle/renderer_ref.py, PyTorch CPU fp32, bit-identical to the reference path on the golden vectors)"}
        result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(finalize(result)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
