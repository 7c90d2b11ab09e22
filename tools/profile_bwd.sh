#!/bin/bash
# Per-kernel durations and PMC counters of the backward chain (tools/bwd_bench.py) on the GPU box.
#   tools/profile_bwd.sh TAG [batch]
set -u
TAG=${1:-r1}; B=${2:-1}
R=$PWD; OUT=$R/gpurun_out/bwdprof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python $R/tools/bwd_bench.py $B 20 > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- python $R/tools/bwd_bench.py $B 5 > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
st = glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True)
with open(os.path.join(out, "summary.txt"), "w") as f:
    if st:
        f.write(f"{'kernel':<80} {'calls':>6} {'avg_ns':>12} {'pct':>7}\n")
        for r in list(csv.DictReader(open(st[0])))[:8]:
            f.write(f"{r['Name'][:80]:<80} {r['Calls']:>6} {float(r['AverageNs']):>12.0f} {r['Percentage']:>7}\n")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        c = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
        if not c:
            f.write(f"{os.path.basename(d)}: no counter csv\n"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(c[0])):
            a = agg[r.get('Kernel_Name', '?')[:60]][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
        for k, cs in agg.items():
            if 'siren_bwd' in k or 'bwd_reduce' in k or 'composite' in k:
                f.write(f"{k}\n")
                for cn, (tot, n) in sorted(cs.items()):
                    f.write(f"    {cn:<32} mean/dispatch = {tot / max(n, 1):.6g}   (n={n})\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
find "$OUT" -type f -size +2M -delete
