#!/bin/bash
# Round 6: A/B of the backward-type kernel generations on one box (tools/bwd_ab.py) at S = 24 and S = 18, the strict-wait variant of the
# 8-wave kernels as a race check, then the backward tests.   bash tools/r6_bwd_ab.sh [quick]
mkdir -p gpurun_out
export BWD_AB_OUT=gpurun_out/r6_bwd_ab_s24.json
timeout 600 python tools/bwd_ab.py 1 20 2>&1 | tee gpurun_out/r6_bwd_ab_s24.txt
export BWD_AB_S=18 BWD_AB_OUT=gpurun_out/r6_bwd_ab_s18.json
timeout 600 python tools/bwd_ab.py 1 20 2>&1 | tee gpurun_out/r6_bwd_ab_s18.txt
if [ "$1" != quick ]; then
  bash tools/build_variant.sh t3strict -DE3DGE_T3_STRICT > gpurun_out/r6_build_strict.log 2>&1
  E3DGE_LIB_PATH=cvpr23-e3dge_amd/lib/variants/lib_t3strict.so BWD_AB_MODES=f32,f16x3_g2 BWD_AB_OUT=gpurun_out/r6_bwd_ab_s18_strict.json \
    timeout 600 python tools/bwd_ab.py 1 20 2>&1 | tee gpurun_out/r6_bwd_ab_s18_strict.txt
fi
timeout 1500 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -15 | tee gpurun_out/r6_test_backward.txt
