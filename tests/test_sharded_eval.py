"""CPU, world_size 2 over gloo: the multi-GPU path of SURVEY.md 8e (image sharding + one metrics all-gather) is
exercised with real processes; the per-unit work is the oracle's renderer on tiny images (tests may use it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import e3dge_amd  # noqa: F401
from e3dge_amd import sharded_eval as se


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 5, 8, 2824):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in se.shard_indices(n, r, w))
            assert got == list(range(n))
            sizes = [len(se.shard_indices(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        se.shard_indices(4, 2, 2)


def test_single_process_gather_is_identity():
    rows = torch.arange(24, dtype=torch.float32).reshape(3, 8)
    out = se.gather_metric_rows(rows, 3, 0, 1)
    assert torch.equal(out, rows)


def test_forced_collective_on_a_single_rank_gloo():
    """force_collective=True takes the all_gather branch even for one rank (the GPU twin of this test runs it over nccl = RCCL)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rows = torch.arange(40, dtype=torch.float32).reshape(5, 8)
        out = se.gather_metric_rows(rows, 5, 0, 1, force_collective=True)
        assert torch.equal(out, rows)
        table = se.evaluate_sharded(lambda i: torch.full((8,), float(i)), 3, 0, 1, force_collective=True)
        assert torch.equal(table[:, 0], torch.arange(3.0))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
        from conftest import full_state_dict
        from e3dge_amd import synthetic as syn
        from oracle import camera_ref, renderer_ref
        torch.set_num_threads(2)
        sd = full_state_dict()[1]
        poses, focal, near, far = camera_ref.camera_from_locations(4, torch.zeros(1, 2))

        def unit(i):     # "render image i, score it": 8 scalars derived from the oracle's render of styles seed i
            wr, _ = syn.synthetic_inputs(1, seed=100 + i)
            with torch.no_grad():
                o = renderer_ref.render(sd, poses, focal, near, far, wr, res=4, n_samples=16)
            img, dep = o['gen_thumb_imgs'], o['depth']
            return torch.stack([img.pow(2).mean(), img.abs().mean(), img.mean(), dep.mean(), dep.min(), dep.max(),
                                o['features'].abs().mean(), torch.tensor(float(i))])
        table = se.evaluate_sharded(unit, n_units, rank, world)
        np.save(os.path.join(out_dir, f"table_{rank}.npy"), table.numpy())
        if rank == 0:
            ref = torch.stack([unit(i) for i in range(n_units)])
            np.save(os.path.join(out_dir, "serial.npy"), ref.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [5, 4])      # ragged (3 + 2) and even (2 + 2) shards
def test_two_rank_sharded_eval_matches_serial(tmp_path, n_units):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_units, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "table_0.npy"), np.load(tmp_path / "table_1.npy")
    serial = np.load(tmp_path / "serial.npy")
    assert t0.shape == (n_units, 8)
    np.testing.assert_array_equal(t0, t1)                      # every rank holds the full table
    np.testing.assert_array_equal(t0[:, 7], np.arange(n_units))  # global unit order restored
    np.testing.assert_allclose(t0, serial, rtol=0, atol=0)     # same process-local arithmetic -> identical rows
    assert not np.isnan(t0).any()                              # padding rows never leak


def test_bench_self_launches_the_ranks():
    """`python bench.py --gpus 2` without torchrun must start 2 ranks itself: the launcher + rendezvous path is
    exercised here on CPU with --dry-run (gloo, no kernel work); on a GPU box the same path runs the bench over RCCL."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["ranks_joined"] == 2 and j["self_launched"] is True
    # the work split bench.py --gpus N will use: C3 at the evaluation set's size, C4 poses k -> rank k mod W,
    # the stage-1 step on every rank with the encoder's 1.03 GB gradient all-reduce, one core set per rank
    assert j["c3"]["images"] == 2824 and j["c3"]["images_per_rank"] == [1412, 1412]
    assert j["c4"] == {"poses": 120, "poses_per_rank": [60, 60]}
    assert j["train_step"] == {"allreduce_bytes": 1.03e9, "ranks": 2}
    assert len(j["cores_per_rank"]) == 2 and all(c >= 1 for c in j["cores_per_rank"]) and j["host_threads"] >= 1
    # a world size that disagrees with --gpus is refused instead of silently running one rank
    bad = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_image_metrics_columns():
    """The 8 columns of builder.py:174-184 on a known pair: identical images -> l2 0, SSIM 1; PSNR formula; SSIM drops with noise."""
    g = torch.Generator().manual_seed(0)
    a = torch.tanh(torch.randn(1, 3, 32, 32, generator=g))
    m = se.image_metrics(a, a + 0.1)
    assert m.shape == (8,)
    np.testing.assert_allclose(float(m[0]), 0.01, rtol=1e-5)                       # loss_l2
    np.testing.assert_allclose(float(m[4]), 0.1, rtol=1e-5)                        # mae
    np.testing.assert_allclose(float(m[5]), 10 * np.log10(1 / 0.0025), rtol=1e-4)   # PSNR on [0,1]-scaled images
    assert float(m[1]) == 0 and float(m[2]) == 0 and float(m[7]) == 1               # pretrained-net terms
    same = se.image_metrics(a, a.clone())
    assert float(same[0]) == 0 and abs(float(same[6]) - 1) < 1e-6
    noisy = se.image_metrics(a, a + 0.3 * torch.randn(a.shape, generator=g))
    assert float(noisy[6]) < 0.95 and float(m[6]) < 1
