#!/bin/bash
# tools/build_variant_one.sh <name> <source stem> [-DFLAG=..]...  -> cvpr23-e3dge_amd/lib/variants/lib_<name>.so
# Like build_variant.sh, but recompiles ONE source with the flags and links it against the default build's objects of the others
# (cvpr23-e3dge_amd/lib/*.o travel with the snapshot): seconds instead of minutes per A/B variant.
set -e
NAME=$1; STEM=$2; shift; shift
D=cvpr23-e3dge_amd
mkdir -p $D/lib/variants
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -Wno-unused-result"
hipcc $FL "$@" -c $D/csrc/$STEM.hip -o $D/lib/variants/${NAME}_$STEM.o
OBJS=""
for f in stream_ops upfirdn2d siren siren_bwd resblock modconv decoder2 local_query metrics align_volume hitprob siren_ws wgrad; do
  if [ $f = $STEM ]; then OBJS="$OBJS $D/lib/variants/${NAME}_$STEM.o"; else OBJS="$OBJS $D/lib/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $D/lib/variants/lib_$NAME.so
rm -f $D/lib/variants/${NAME}_$STEM.o
echo built $D/lib/variants/lib_$NAME.so
