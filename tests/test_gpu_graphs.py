"""GPU: HIP-graph replay of a fixed launch sequence (e3dge_amd.graphs.GraphedCall) -- the renderer + decoder forward is
capturable (no host synchronisation, no allocation outside torch's allocator) and replays bit-identically."""
import pytest
import torch

from conftest import full_state_dict

import e3dge_amd  # noqa: F401
from e3dge_amd import synthetic as syn
from e3dge_amd.camera_utils import generate_camera_params
from e3dge_amd.graphs import GraphedCall

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_generator_forward_replays_bit_identically_with_new_inputs():
    g, _ = full_state_dict(size=128, cm=1, res=32, n_samples=24)
    g = g.to(DEV).eval()
    g.requires_grad_(False)
    poses, focal, near, far, _ = generate_camera_params(32, DEV, locations=torch.zeros(1, 2, device=DEV))
    codes = [syn.synthetic_inputs(1, seed=s, device=DEV) for s in (1, 2, 3)]
    codes = [(wr, wd[:, :g.decoder.n_latent].contiguous()) for wr, wd in codes]

    def fwd(wr, wd):
        o = g([wr, wd], poses, focal, near, far, input_is_latent=True, randomize_noise=False)
        return o['gen_imgs'], o['gen_thumb_imgs'], o['depth']
    with torch.no_grad():
        eager = [[t.clone() for t in fwd(*c)] for c in codes]
        gc = GraphedCall(fwd, *codes[0])
        for c, want in zip(codes + codes[:1], eager + eager[:1]):
            got = gc(*c)
            torch.cuda.synchronize()
            for a, b in zip(got, want):
                assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        gc(codes[0][0])
    with pytest.raises(RuntimeError):
        GraphedCall(fwd, torch.zeros(3))
