"""The RCCL legs of the multi-GPU path, executed on the one GPU a lease has: a process group of world size 1 over
`nccl` (= RCCL on ROCm).  It proves nothing about scaling; it proves that the collectives this path issues -- the metrics
all_gather of sharded_eval.gather_metric_rows (reference: the loss-vector reduce of project/utils/dist_utils.py:108-130) and the
emulated 1.03 GB encoder-gradient all-reduce bench.py overlaps with a C5 step (trainer.py:1737-1778) -- run on device tensors, on
the streams the path uses, without changing a bit of the step's result."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist

import e3dge_amd  # noqa: F401
from e3dge_amd import sharded_eval as se, synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from conftest import full_state_dict, record
from test_gpu_renderer import make_renderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        yield dist.group.WORLD
    finally:
        dist.destroy_process_group()


def test_metrics_all_gather_over_rccl(rccl_group):
    """gather_metric_rows(..., force_collective=True): a real dist.all_gather of device rows; evaluate_sharded with the HIP metric
    kernel as the unit."""
    assert dist.get_backend() == "nccl"
    rows = torch.arange(5 * 8, dtype=torch.float32, device=DEV).reshape(5, 8)
    out = se.gather_metric_rows(rows, 5, 0, 1, force_collective=True)
    torch.cuda.synchronize()
    assert out.device.type == "cuda" and torch.equal(out, rows)
    g = torch.Generator(device=DEV).manual_seed(0)
    imgs = [torch.rand(1, 3, 64, 64, device=DEV, generator=g) * 2 - 1 for _ in range(4)]
    gt = torch.rand(1, 3, 64, 64, device=DEV, generator=g) * 2 - 1
    table = se.evaluate_sharded(lambda i: se.image_metrics(imgs[i], gt), 4, 0, 1, device=DEV, force_collective=True)
    plain = se.evaluate_sharded(lambda i: se.image_metrics(imgs[i], gt), 4, 0, 1, device=DEV)
    assert torch.equal(table, plain) and not torch.isnan(table).any()


def test_emulated_encoder_all_reduce_beside_a_c5_step(rccl_group):
    """bench.py's C5 leg at N > 1: a 1.03 GB fp32 bucket all-reduced on a side stream while the renderer's training step runs
    (DDP overlaps the encoder's gradient all-reduce with backward).  Same gradient, bit for bit, with and without it."""
    res, S = 16, 18
    r = make_renderer(full_state_dict(res=res, n_samples=S)[1], res, S)
    wr, _ = syn.synthetic_inputs(1, seed=5, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, locations=torch.zeros(1, 2, device=DEV))

    def step():
        s_ = wr.clone().requires_grad_(True)
        o = r(poses, focal, near, far, styles=s_, return_eikonal=True, return_surface_eikonal=True)
        ((o['gen_thumb_imgs'] ** 2).mean() + ((o['eikonal_term'].norm(dim=-1) - 1) ** 2).mean()
         + (o['surface_eikonal_term'] ** 2).mean()).backward()
        return s_.grad
    alone = step().clone()
    bucket = torch.ones(int(1.03e9 // 4), device=DEV)
    side = torch.cuda.Stream(device=DEV)

    def both():
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            dist.all_reduce(bucket)
        g = step()
        torch.cuda.current_stream(DEV).wait_stream(side)
        return g
    for _ in range(2):
        g = both()
    torch.cuda.synchronize()
    assert torch.equal(g, alone)
    assert float(bucket[0]) == 1.0 and float(bucket[-1]) == 1.0          # sum over one rank
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        dist.all_reduce(bucket)
    side.synchronize()
    record("rccl_world1_allreduce_1p03GB", allreduce_ms=1e3 * (time.perf_counter() - t0))
