import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: imports /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    has_ref = os.path.isdir("/root/reference/project")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU visible"))
        if "needs_reference" in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand with hipcc; cross-compiles without a GPU)."""
    import e3dge_amd  # noqa: F401
    from e3dge_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def full_state_dict(size=256, cm=1, res=64, n_samples=24):
    """Synthetic weights keyed exactly like the reference's g_ema state dict (same values as the fixtures)."""
    import e3dge_amd  # noqa: F401
    from e3dge_amd import synthetic as syn
    from e3dge_amd.stylesdf_model import G_pred_latents
    g = G_pred_latents(syn.model_opt(size=size, channel_multiplier=cm, renderer_spatial_output_dim=res),
                       syn.rendering_opt(N_samples=n_samples), full_pipeline=True)
    syn.load_synthetic(g)
    return g, {k: v.clone() for k, v in g.state_dict().items()}


def record(name, **vals):
    """Append a line to gpurun_out/parity_report.jsonl (merged back from the GPU box) -- measured errors are
    kept even when an assertion fails later."""
    import json
    d = os.path.join(REPO, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **{k: (float(v) if hasattr(v, '__float__') else v) for k, v in vals.items()})) + "\n")


def maxerr(a, b):
    import torch
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())


def contraction_modes(*modes):
    """Contraction modes to parametrize over: the experimental one (first-generation split-f16 forward 'f16x3_v1') exists only in
    -DE3DGE_EXPERIMENTAL builds of the library (include/e3dge_hip_experimental.h) and is dropped when the loaded library does not
    have it.  'f16x3_g2' (the 8-wave backward-type kernels, csrc/siren16_bwd.h) is part of every build since round 6."""
    import e3dge_amd  # noqa: F401
    from e3dge_amd import _lib
    try:
        exp = _lib.has_experimental()
    except Exception:
        exp = False
    return [m for m in modes if exp or m != "f16x3_v1"]
