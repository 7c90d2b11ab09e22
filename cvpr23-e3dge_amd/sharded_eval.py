"""Image-sharded evaluation across GPUs (BASELINE.json configs[2] and [3]).

The reference evaluates on ONE GPU (`scripts/test/eval_2dmetrics_ffhq.sh:24-26`, `trainer.py:290-556`); its only
collective anywhere near this path is the loss-vector reduce of `dist_utils.py:108-130`.  Every image (or camera
pose of a sweep) is an independent unit, so the MI355X design is: one process per GPU, unit i -> rank i mod W,
weights replicated, NO collective inside the path, and exactly one RCCL `all_gather` of the per-unit metric
rows at the end (ragged tails padded).  The 8 columns are the scalars `losses/builder.py:174-184` reports:
loss_l2, loss_id, loss_lpips, loss, mae, PSNR, SSIM, ID_SIM -- whatever `unit_fn` returns is gathered verbatim.

`backend` is "nccl" (= RCCL on ROCm) on GPUs and "gloo" in the CPU tests; the logic is identical."""
import torch
import torch.distributed as dist

N_METRICS = 8


def shard_indices(n_units, rank, world_size):
    """Units owned by `rank`: i = rank, rank + W, rank + 2W, ...  (interleaved, so ranks stay balanced when the
    cost of a unit drifts along the list, e.g. along a camera sweep)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_units, world_size))


def gather_metric_rows(local_rows, n_units, rank, world_size, group=None, force_collective=False):
    """local_rows (n_local, M) on this rank's device, rows in shard_indices order -> (n_units, M) in global unit
    order on EVERY rank.  One all_gather of a (ceil(n/W), M) block per rank; ragged tails are padded with NaN and
    dropped after the gather.  A single rank needs no collective; `force_collective=True` issues the all_gather anyway
    (an initialised process group is then required): with it a one-GPU lease executes the RCCL leg of this path."""
    n_local = len(shard_indices(n_units, rank, world_size))
    if local_rows.ndim != 2 or local_rows.shape[0] != n_local:
        raise ValueError(f"expected ({n_local}, M) rows on rank {rank}, got {tuple(local_rows.shape)}")
    per_rank = (n_units + world_size - 1) // world_size
    M = local_rows.shape[1]
    block = torch.full((per_rank, M), float('nan'), dtype=local_rows.dtype, device=local_rows.device)
    block[:n_local] = local_rows
    if world_size == 1 and not force_collective:
        gathered = [block]
    else:
        gathered = [torch.empty_like(block) for _ in range(world_size)]
        dist.all_gather(gathered, block, group=group)
    out = torch.empty((n_units, M), dtype=local_rows.dtype, device=local_rows.device)
    for r in range(world_size):
        idx = shard_indices(n_units, r, world_size)
        if idx:
            out[torch.tensor(idx, device=out.device)] = gathered[r][:len(idx)]
    return out


def evaluate_sharded(unit_fn, n_units, rank=0, world_size=1, device="cpu", group=None, n_streams=1, force_collective=False):
    """Runs `unit_fn(i) -> (M,) tensor of metric scalars` for the units this rank owns and returns the (n_units, M)
    table (identical on all ranks).  `unit_fn` is where the hot path is called (renderer / generator forward on
    this rank's GPU); nothing is exchanged between ranks until the final gather.
    n_streams > 1 (GPU only): the rank's units are issued round-robin on that many HIP streams -- units are independent, and
    one unit's small launches (the decoder's first levels, metric folds) leave CUs idle that the next unit's kernels fill."""
    mine = list(shard_indices(n_units, rank, world_size))
    if n_streams > 1 and torch.device(device).type == "cuda" and mine:
        cur = torch.cuda.current_stream(device)
        streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        rows = [unit_fn(mine[0]).reshape(-1).to(device)]       # first unit on the caller's stream: lazily built weight images
        for st in streams:                                     # (packed once, then read-only) exist before the fan-out
            st.wait_stream(cur)
        for j, i in enumerate(mine[1:]):
            with torch.cuda.stream(streams[j % n_streams]):
                row = unit_fn(i).reshape(-1).to(device)
            row.record_stream(cur)
            rows.append(row)
        for st in streams:
            cur.wait_stream(st)
    else:
        rows = [unit_fn(i).reshape(-1).to(device) for i in mine]
    if rows:
        local = torch.stack(rows)
    else:
        probe = unit_fn.__dict__.get('n_metrics', N_METRICS)
        local = torch.empty((0, probe), device=device)
    return gather_metric_rows(local, n_units, rank, world_size, group, force_collective)


# ------------------------------------------------------------------------------------------------------------------
# the 8 metric columns of one evaluated image (losses/builder.py:130-184)
# ------------------------------------------------------------------------------------------------------------------
def _gaussian_window(size, sigma, device, dtype):
    x = torch.arange(size, device=device, dtype=dtype) - (size - 1) / 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim_index(x, y, window=5, max_val=1.0):
    """Mean structural similarity with a Gaussian window (sigma 1.5, reflect padding) -- the quantity behind the
    reference's `1 - kornia.losses.ssim_loss(pred, gt, 5)` (builder.py:166,181): ssim_loss = mean(clamp((1 - ssim)/2, 0, 1)),
    so the reported SSIM column is 1 - that."""
    import torch.nn.functional as F
    C = x.shape[1]
    w = _gaussian_window(window, 1.5, x.device, x.dtype).expand(C, 1, window, window).contiguous()
    pad = window // 2
    f = lambda t: F.conv2d(F.pad(t, [pad] * 4, mode='reflect'), w, groups=C)
    mx, my = f(x), f(y)
    sxx, syy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    ssim = ((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))
    return 1 - torch.clamp((1 - ssim) / 2, 0, 1).mean()


def image_metrics(pred, gt, l2_lambda=1.0):
    """The 8 metric columns of one image pair (see image_metrics_torch).  GPU fp32 tensors take the fused HIP kernel
    e3dge_image_metrics (one pass over both images); anything else the plain-torch formulation."""
    if pred.device.type == "cuda" and pred.dtype == torch.float32 and gt.dtype == torch.float32 and pred.ndim == 4 \
            and pred.shape == gt.shape and not (torch.is_grad_enabled() and (pred.requires_grad or gt.requires_grad)):
        from . import _lib
        lib = _lib.load()
        B, C, H, W = pred.shape
        p, g = pred.contiguous(), gt.contiguous()
        sums = torch.empty((B, 4), device=p.device, dtype=torch.float32)
        scratch = torch.empty(lib.e3dge_image_metrics_scratch_floats(B, C, H, W), device=p.device, dtype=torch.float32)
        with torch.cuda.device(p.device):
            rc = lib.e3dge_image_metrics(_lib.ptr(sums), _lib.ptr(scratch), _lib.ptr(p), _lib.ptr(g), B, C, H, W, 1.0,
                                         _lib.stream_of(p))
            _lib.check(rc, "e3dge_image_metrics")
            # the reference's losses are means over the whole batch tensor; the eight columns in one more launch (as torch ops
            # this tail was thirteen 5-us kernels: 3 % of an evaluated image)
            row = torch.empty(8, device=p.device, dtype=torch.float32)
            rc = lib.e3dge_image_metric_row(_lib.ptr(row), _lib.ptr(sums), B, float(l2_lambda), _lib.stream_of(p))
        _lib.check(rc, "e3dge_image_metric_row")
        return row
    return image_metrics_torch(pred, gt, l2_lambda)


def image_metrics_torch(pred, gt, l2_lambda=1.0):
    """(8,) = [loss_l2, loss_id, loss_lpips, loss, mae, PSNR, SSIM, ID_SIM] for one predicted image against its target,
    both (1,3,H,W) in [-1,1] (calc_2d_rec_loss, builder.py:130-184).  The two terms that need pretrained networks the
    image does not have (ArcFace identity, LPIPS/VGG) are outside the hot path and reported as 0 (ID_SIM = 1 - 0), exactly
    as the reference does when their lambdas are 0 (:145, :158-163)."""
    mse = torch.mean((pred - gt) ** 2)
    zero = torch.zeros((), device=pred.device, dtype=pred.dtype)
    p01, g01 = pred / 2 + 0.5, gt / 2 + 0.5
    psnr = 10.0 * torch.log10(1.0 / torch.mean((p01 - g01) ** 2))
    return torch.stack([mse, zero, zero, mse * l2_lambda, torch.mean((pred - gt).abs()), psnr, ssim_index(pred, gt), 1 - zero])
