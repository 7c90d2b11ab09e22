#!/bin/bash
# Kernel trace + PMC of the decoder's 3x3 layers on the fused modulated-conv kernel (tools/modconv_bench.py hip).
#   tools/profile_modconv.sh r2   -> gpurun_out/prof_modconv_r2/{kernel_stats_summary.txt,pmc_summary.txt}
set -u
TAG=${1:-r2}
OUT=$PWD/gpurun_out/prof_modconv_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/modconv_bench.py hip"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $CMD > "$OUT/pmc_$name.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
st = glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(out, "kernel_stats_summary.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python tools/modconv_bench.py hip   (13 calls per layer)\n")
        f.write(f"{'kernel':<100} {'calls':>6} {'avg_ns':>12} {'pct':>7}\n")
        for r in rows[:24]:
            f.write(f"{r['Name'][:100]:<100} {r['Calls']:>6} {float(r['AverageNs']):>12.0f} {r['Percentage']:>7}\n")
    print(open(os.path.join(out, "kernel_stats_summary.txt")).read())
with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        c = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
        if not c:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(c[0])):
            k = r.get('Kernel_Name', r.get('Kernel Name', '?'))[:80]
            a = agg[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
        for k, cs in agg.items():
            if 'modconv_kernel' in k or 'upfirdn2d_tiled' in k:
                f.write(f"{k}\n")
                for cn, (tot, n) in sorted(cs.items()):
                    f.write(f"    {cn:<32} mean/dispatch = {tot / max(n,1):.6g}   (n={n})\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
PY
find "$OUT" -type f -size +2M -delete
