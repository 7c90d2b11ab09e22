#!/bin/bash
# tools/ws_variant.sh <name> [-DFLAG=..]...  -> cvpr23-e3dge_amd/lib/variants/ws_<name>.so (siren_ws.hip + the error helpers only)
set -e
NAME=$1; shift
D=cvpr23-e3dge_amd
mkdir -p $D/lib/variants
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -Wno-unused-result"
hipcc $FL "$@" -shared $D/csrc/siren_ws.hip $D/csrc/stream_ops.hip -o $D/lib/variants/ws_$NAME.so
echo built $D/lib/variants/ws_$NAME.so
