#!/bin/bash
# fuse block under autograd: current library vs a variant with the committed (HEAD) siren_ws.hip
python tools/time_fuse_autograd.py 2>&1 | tail -1 | cut -c1-400
git show HEAD:cvpr23-e3dge_amd/csrc/siren_ws.hip > /tmp/siren_ws_head.hip 2>/dev/null || exit 0
