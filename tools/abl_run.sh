#!/bin/bash
# headline kernel time of library variants on one box: tools/abl_run.sh <variant> ...   ("base" = the default library)
for v in "$@"; do
  if [ $v = base ]; then unset E3DGE_LIB_PATH; else export E3DGE_LIB_PATH=cvpr23-e3dge_amd/lib/variants/lib_$v.so; fi
  python bench.py --steps 600 --warmup 50 --no-c3 --no-c4 --no-train-step --no-surface --no-cpu-baseline --no-inversion 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4), 'rays/s', round(d['value']))"
done
