#!/bin/bash
# Same PMC counter groups on the fp32 forward kernel (bench.py, E3DGE_MFMA_MODE=f32) and on the backward chain
# (tools/bwd_bench.py), to compare where the wave cycles go.   tools/profile_cmp.sh TAG
set -u
TAG=${1:-cmp}
R=$PWD; OUT=$R/gpurun_out/cmp_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH"
G3="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
i=0
for grp in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  E3DGE_MFMA_MODE=f32 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/fwd_$i" -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inversion --no-train-step > "$OUT/fwd_$i.log" 2>&1; echo "fwd $i rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/bwd_$i" -o pmc -- python $R/tools/bwd_bench.py 1 5 > "$OUT/bwd_$i.log" 2>&1; echo "bwd $i rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for d in sorted(glob.glob(os.path.join(out, "*_[0-9]"))):
        c = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
        if not c:
            f.write(f"{os.path.basename(d)}: no counter csv\n"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(c[0])):
            a = agg[r.get('Kernel_Name', '?')[:60]][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
        for k, cs in agg.items():
            if 'siren_bwd_kernel' in k or 'siren_kernel' in k:
                f.write(f"{os.path.basename(d)} {k}\n")
                for cn, (tot, n) in sorted(cs.items()):
                    f.write(f"    {cn:<32} mean/dispatch = {tot / max(n, 1):.6g}   (n={n})\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
find "$OUT" -type f -size +2M -delete
