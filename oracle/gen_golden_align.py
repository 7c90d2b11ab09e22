"""Golden vectors for align_volume (surface extraction), recorded from the REAL reference function
(project/utils/mesh_utils.py:17-44, imported via oracle/ref_harness.py) -- authoring container only.  TEST INFRASTRUCTURE.

    python oracle/gen_golden_align.py        # writes tests/golden/align_volume.npz (+ align_volume_report.json)

Cases: a ragged single-channel volume (11, 13, 17), a cube (24^3) with the default near/far, and one with a wide frustum
(near 0.5, far 1.5) where most of the far slices leave the unit cube; inputs are recorded by seed."""
import importlib
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")

from oracle import mesh_ref, ref_harness  # noqa: E402

CASES = [dict(name="ragged", shape=(1, 11, 13, 17, 1), near=0.88, far=1.12, seed=1),
         dict(name="cube24", shape=(1, 24, 24, 24, 1), near=0.88, far=1.12, seed=2),
         dict(name="wide", shape=(1, 16, 12, 20, 1), near=0.5, far=1.5, seed=3)]


def case_volume(case):
    rs = np.random.RandomState(case["seed"])
    return torch.from_numpy(rs.normal(size=case["shape"]).astype(np.float32))


def main():
    ref_harness.modules()                                   # installs the stub modules the reference imports need
    mu = importlib.import_module('project.utils.mesh_utils')
    arrays, report = {}, {}
    for case in CASES:
        vol = case_volume(case)
        ref = mu.align_volume(vol.clone(), near=case["near"], far=case["far"])
        mine = mesh_ref.align_volume(vol, near=case["near"], far=case["far"])
        report[case["name"]] = dict(max_abs_diff_restatement_vs_reference=float((ref - mine).abs().max()),
                                    bit_exact=bool(torch.equal(ref, mine)),
                                    outside_fraction=float((ref == 1).float().mean()))
        arrays["ref_" + case["name"]] = ref.numpy()
    print(json.dumps(report, indent=1))
    with open(os.path.join(GOLD, "align_volume_report.json"), "w") as f:
        json.dump(dict(cases=CASES, report=report), f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "align_volume.npz"), **arrays)


if __name__ == "__main__":
    main()
