#!/bin/bash
# round 6: second-generation e3dge_wgrad -- tests, time per size and block shape, the trainable Fuse_sft_MLP / texture head / stage-2 steps
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_texhead.py tests/test_local_query.py tests/test_gpu_stage2.py tests/test_gpu_graphs.py -q -m gpu > $O/r6_wgrad_tests.log 2>&1
grep -E "passed|failed" $O/r6_wgrad_tests.log | tail -2
: > $O/r6_wgrad_shapes.jsonl
for sh in auto 0 1 2; do
  if [ $sh = auto ]; then unset E3DGE_WGRAD_SHAPE; else export E3DGE_WGRAD_SHAPE=$sh; fi
  timeout 200 python tools/time_wgrad.py 2>/dev/null | tail -1 >> $O/r6_wgrad_shapes.jsonl
done
unset E3DGE_WGRAD_SHAPE
cat $O/r6_wgrad_shapes.jsonl
timeout 300 python tools/time_fuse_autograd.py 2>/dev/null | tail -1
timeout 300 python tools/time_texhead_autograd.py 2>/dev/null | tail -1
timeout 300 python tools/stage2_step.py 10 2>/dev/null | tail -1
