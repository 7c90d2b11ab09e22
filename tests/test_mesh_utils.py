"""Surface extraction, device half: align_volume against the reference's own outputs (fixture recorded by
oracle/gen_golden_align.py from project/utils/mesh_utils.py:17-44), the oracle restatement, and the HIP kernel."""
import json
import os

import numpy as np
import pytest
import torch

import e3dge_amd  # noqa: F401
from e3dge_amd import mesh_utils
from oracle import mesh_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "align_volume_report.json")))["cases"]
DEV = "cuda:0"


def case_volume(case):
    rs = np.random.RandomState(case["seed"])
    return torch.from_numpy(rs.normal(size=tuple(case["shape"])).astype(np.float32))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "align_volume.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_and_cpu_branch_reproduce_the_reference_bit_for_bit(gold, case):
    vol = case_volume(case)
    ref = torch.from_numpy(gold["ref_" + case["name"]])
    assert torch.equal(mesh_ref.align_volume(vol, case["near"], case["far"]), ref)
    assert torch.equal(mesh_utils.align_volume(vol, case["near"], case["far"]), ref)
    assert (ref == 1).float().mean() > 0.2                     # the frustum really cuts voxels away in every case


def test_cpu_shape_and_type_errors():
    with pytest.raises(RuntimeError):
        mesh_utils.align_volume(torch.zeros(4, 4, 4))
    out = mesh_utils.align_volume(torch.zeros(2, 3, 4, 5, 2))
    assert out.shape == (2, 3, 4, 5, 2)


def test_marching_cubes_reports_missing_third_party_packages():
    try:
        import skimage  # noqa: F401
        import trimesh  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="outside the accelerated path"):
            mesh_utils.marching_cubes_mesh(torch.zeros(1, 4, 4, 4, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_kernel_against_the_reference_fixture(gold, case):
    vol = case_volume(case).to(DEV)
    out = mesh_utils.align_volume(vol, case["near"], case["far"]).cpu()
    ref = torch.from_numpy(gold["ref_" + case["name"]])
    assert torch.equal(out, ref), float((out - ref).abs().max())          # same operations in the same order: bit-exact


@pytest.mark.gpu
def test_hip_kernel_full_size_batch_and_channels_against_the_oracle():
    rs = np.random.RandomState(11)
    vol = torch.from_numpy(rs.normal(size=(1, 128, 128, 128, 1)).astype(np.float32))
    out = mesh_utils.align_volume(vol.to(DEV)).cpu()
    assert torch.equal(out, mesh_ref.align_volume(vol))
    # the last depth slice (coef = 1) samples the grid points themselves, up to the rounding of the index arithmetic;
    # the outer frustum is 1
    assert torch.allclose(out[0, 1:-1, 1:-1, -1, 0], vol[0, 1:-1, 1:-1, -1, 0], atol=1e-4)
    assert float(out[0, 0, 0, 0, 0]) == 1.0
    vol2 = torch.from_numpy(rs.normal(size=(2, 9, 12, 7, 3)).astype(np.float32))
    assert torch.equal(mesh_utils.align_volume(vol2.to(DEV), 0.8, 1.3).cpu(), mesh_ref.align_volume(vol2, 0.8, 1.3))
    with pytest.raises(RuntimeError):
        mesh_utils.align_volume(vol2.double().to(DEV))


@pytest.mark.gpu
def test_renderer_return_mesh_returns_the_aligned_volume():
    from e3dge_amd import synthetic as syn
    from e3dge_amd.camera_utils import generate_camera_params
    from e3dge_amd.volume_renderer import VolumeFeatureRenderer
    res = 16                                                   # surf_extraction: N_samples = renderer output size (train_setup.py:112-126)
    r = VolumeFeatureRenderer(syn.rendering_opt(N_samples=res), out_im_res=res, mode='test')
    syn.load_synthetic(r, prefix='renderer.')
    r = r.to(DEV)
    wr, _ = syn.synthetic_inputs(1, seed=4, device=DEV)
    poses, focal, near, far, _ = generate_camera_params(res, DEV, batch=1)
    with torch.no_grad():
        out = r(poses, focal, near, far, styles=wr, return_mesh=True)
    assert out['aligned_sdf'].shape == (1, res, res, res, 1)
    assert torch.equal(out['aligned_sdf'].cpu(), mesh_ref.align_volume(out['sdf'].cpu()))
    assert 'mesh' in out
