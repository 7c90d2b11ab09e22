for v in base abl8 abl16 abl32 abl56 base; do
  if [ $v = base ]; then unset E3DGE_LIB_PATH; else export E3DGE_LIB_PATH=cvpr23-e3dge_amd/lib/variants/lib_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-c3 --no-c4 --no-train-step --no-surface --no-cpu-baseline --no-inversion 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4))"
done
