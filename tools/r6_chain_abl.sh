#!/bin/bash
# Round 6: A/B of the 8-wave training kernels' build options by timing (tools/bwd_ab.py, S = 18): issue roles, slab-major layout of the
# saved state (timing only while the forward still writes point-major arguments), ablations (E3DGE_T3_ABL bits, siren16_bwd.h).
#   bash tools/r6_chain_abl.sh name:flags ...      e.g.  base:  nosplit:-DE3DGE_T3_SPLIT=0  "blocked:-DE3DGE_T3_BLOCKED=1"
mkdir -p gpurun_out
export BWD_AB_S=${BWD_AB_S:-18} BWD_AB_MODES=${BWD_AB_MODES:-f32,f16x3_g2}
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  if [ $n = base ]; then unset E3DGE_LIB_PATH; else
    bash tools/build_variant_one.sh t3_$n siren_bwd $fl > gpurun_out/r6_abl_build_$n.log 2>&1 || { echo "build of $n failed"; tail -5 gpurun_out/r6_abl_build_$n.log; continue; }
    export E3DGE_LIB_PATH=cvpr23-e3dge_amd/lib/variants/lib_t3_$n.so
  fi
  echo "== $n ($fl)"
  BWD_AB_OUT=gpurun_out/r6_abl_$n.json timeout 300 python tools/bwd_ab.py 1 20 2>&1 | grep "f16x3" | sed 's/, "finite.*//; s/"rel_dev_vs_f32": //'
done 2>&1 | tee -a gpurun_out/r6_chain_abl.txt
