#!/bin/bash
# Profiles bench.py on the GPU box: per-kernel timing (--kernel-trace --stats) and, in SEPARATE runs, PMC
# counters (MFMA busy, VALU, wave cycles; FETCH_SIZE; WRITE_SIZE).  Output under gpurun_out/prof_$TAG/.
#   tools/profile_bench.sh r1 [extra bench args]
set -u
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# only the headline workload in the profiled process: the informational legs launch the same kernels at other sizes
BENCH="python $PWD/bench.py --steps 30 --warmup 5 --headline-only $*"
cd /tmp
echo "== kernel trace" 
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
echo "rc=$?"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  echo "== pmc $grp"
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1
  echo "rc=$?"
done
# compact summaries for profiles/
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None
st = find("trace/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(out, "kernel_stats_summary.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --headline-only\n")
        f.write(f"{'kernel':<90} {'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'pct':>7}\n")
        for r in rows[:40]:
            f.write(f"{r['Name'][:90]:<90} {r['Calls']:>6} {r['TotalDurationNs']:>14} {float(r['AverageNs']):>12.0f} {r['Percentage']:>7}\n")
    print(open(os.path.join(out, "kernel_stats_summary.txt")).read())
with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        c = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
        if not c:
            f.write(f"{os.path.basename(d)}: no counter csv\n"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(c[0])):
            k = r.get('Kernel_Name', r.get('Kernel Name', '?'))[:70]
            a = agg[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
        for k, cs in agg.items():
            if 'siren_kernel' in k or 'siren16_kernel' in k or 'upfirdn' in k or 'bias_act' in k or 'film' in k:
                f.write(f"{k}\n")
                for cn, (tot, n) in sorted(cs.items()):
                    f.write(f"    {cn:<32} mean/dispatch = {tot / max(n,1):.6g}   (n={n})\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
# HBM traffic of the render kernel for bench.py's roofline.traffic (profiles/traffic_pmc.json is assembled from these)
import json, re
vals = {}
txt = open(os.path.join(out, "pmc_summary.txt")).read()
for blk in re.split(r"\n(?=\S)", txt):
    if "siren_kernel<0" in blk or "siren16_kernel<0" in blk:
        for m in re.finditer(r"(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch = ([0-9.e+]+)", blk):
            vals[m.group(1) + "_KB"] = float(m.group(2))
json.dump(vals, open(os.path.join(out, "traffic.json"), "w"))
print("traffic:", vals)
PY
( cd "$OUT" && find . -type f | head -60 > "$OUT/files.txt"; cat "$OUT/files.txt" )
# keep the merge-back small: drop anything above 2 MB (raw traces, databases)
find "$OUT" -type f -size +2M -delete
du -sh "$OUT"
