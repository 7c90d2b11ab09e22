"""GPU: the fused per-image metrics kernel (e3dge_image_metrics) against the plain-torch formulation of the same columns
(sharded_eval.image_metrics_torch: MSE / L1 / PSNR / SSIM as losses/builder.py:130-184 defines them).  Tolerance: the sums
run over up to 3M elements in a different association order: 2e-5 relative per column."""
import numpy as np
import pytest
import torch

from conftest import record

import e3dge_amd  # noqa: F401
from e3dge_amd import sharded_eval as se

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(1, 3, 1024, 1024), (2, 3, 50, 70), (1, 1, 5, 33), (1, 3, 256, 256)])
def test_image_metrics_kernel_vs_torch(shape):
    g = torch.Generator(device=DEV).manual_seed(shape[2])
    gt = torch.tanh(torch.randn(shape, device=DEV, generator=g))
    pred = torch.tanh(gt * 1.2 + 0.1 * torch.randn(shape, device=DEV, generator=g))
    a = se.image_metrics(pred, gt)
    b = se.image_metrics_torch(pred, gt)
    rel = float(((a - b).abs() / b.abs().clamp_min(1e-6)).max())
    record(f"image_metrics_{shape[2]}x{shape[3]}", max_rel_diff=rel, ssim=float(a[6]), psnr=float(a[5]))
    assert a.shape == (8,) and rel <= 2e-5, (a, b)
    same = se.image_metrics(gt, gt.clone())
    assert float(same[0]) == 0.0 and abs(float(same[6]) - 1) < 1e-6
    assert torch.equal(se.image_metrics(pred, gt), a)               # fixed-order fold: bit-reproducible
    half = se.image_metrics(pred, gt, l2_lambda=0.5)                # the row kernel (e3dge_image_metric_row) applies the weight
    assert float(half[3]) == 0.5 * float(a[0]) and torch.equal(half[[0, 4, 5, 6]], a[[0, 4, 5, 6]])
    assert float(a[1]) == 0.0 and float(a[2]) == 0.0 and float(a[7]) == 1.0
