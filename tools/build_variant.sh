#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...  -> cvpr23-e3dge_amd/lib/variants/lib_<name>.so  (kernel A/B experiments)
set -e
NAME=$1; shift
D=cvpr23-e3dge_amd
mkdir -p $D/lib/variants
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -Wno-unused-result"
for f in stream_ops upfirdn2d siren siren_bwd resblock modconv decoder2 local_query metrics align_volume hitprob siren_ws wgrad; do hipcc $FL "$@" -c $D/csrc/$f.hip -o $D/lib/variants/${NAME}_$f.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC $D/lib/variants/${NAME}_*.o -o $D/lib/variants/lib_$NAME.so
rm -f $D/lib/variants/${NAME}_*.o
echo built $D/lib/variants/lib_$NAME.so
