"""Golden vectors for the local-feature -> texture-FiLM head, recorded from the REAL reference class
(project/models/helper_modules/resnetfc.py ResnetBlockFC, imported via oracle/ref_harness.py) -- authoring container
only.  TEST INFRASTRUCTURE.

    python oracle/gen_golden_texhead.py    # writes tests/golden/texhead_301.npz

Weights: cvpr23-e3dge_amd/synthetic.py values for the keys renderer.network.netLocal.local_feat_to_tex_modulations_linear.*
(the reference zero-initialises the block, which would test nothing)."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import e3dge_amd  # noqa: E402,F401
from e3dge_amd import synthetic as syn  # noqa: E402
from oracle import ref_harness, renderer_ref  # noqa: E402
from oracle.gen_golden import maxdiff, npf, save  # noqa: E402

PREFIX = 'renderer.network.netLocal.local_feat_to_tex_modulations_linear.'


def main():
    ref_harness.prepare()
    rfc = importlib.import_module('project.models.helper_modules.resnetfc')
    cin = 301
    blk = rfc.ResnetBlockFC(cin, 512)
    sd = {k: syn.synthetic_tensor(PREFIX + k, v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    feats = syn.synthetic_local_feats(1, 4, 19, cin=cin, seed=5).reshape(1, 4, 4, 19, cin)      # 304 points
    with torch.no_grad():
        out = blk(feats)
        ra, rb = torch.split(out, 256, dim=-1)          # SirenLocalGlobal.forward_backbone :334-336
        full = {PREFIX + k: v for k, v in sd.items()}
        ma, mb = renderer_ref.tex_modulations(full, PREFIX, feats)
        ta, tb = renderer_ref.tex_modulations(full, PREFIX, feats, dtype=torch.float64)
    print(f"  texhead: restatement vs reference max|d| = {max(maxdiff(ra, ma), maxdiff(rb, mb)):.3e}; "
          f"reference vs f64 = {max(maxdiff(ra, ta), maxdiff(rb, tb)):.3e}; |out| max = {float(out.abs().max()):.2f}")
    save("texhead_301", cin=np.int32(cin), feats_seed=np.int32(5), feats_shape=np.int32([1, 4, 4, 19, cin]),
         ref_alpha=npf(ra), ref_beta=npf(rb), f64_alpha=npf(ta), f64_beta=npf(tb))


if __name__ == "__main__":
    main()
