#!/bin/bash
# Stage-1 renderer step at 1, 2 and 4 samples per step: event-timed, then rocprofv3 per-kernel averages (does a kernel's duration scale with
# the number of 128-point tiles, or with the number of 256-CU rounds?).   -> gpurun_out/c5_batch/
set -u
OUT=$PWD/gpurun_out/c5_batch
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for B in 1 2 4; do
  python $REPO/tools/c5_step.py 20 $B 2>&1 | tail -1 | tee -a "$OUT/timed.txt"
done
for B in 1 4; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/b$B" -o t -- python $REPO/tools/c5_step.py 10 $B > "$OUT/b$B.log" 2>&1
  python - "$OUT/b$B" "$OUT/kernel_stats_b$B.txt" $B <<'PY'
# per (kernel, grid) averages: the ray samples' launch and the surface points' launch of the same kernel are different rows
import collections, csv, glob, os, re, sys
d, out, B = sys.argv[1], sys.argv[2], sys.argv[3]
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(tr[0])):
    if "e3dge::" not in r["Kernel_Name"]:
        continue
    name = re.sub(r"void |e3dge::|\(.*", "", r["Kernel_Name"])
    wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    a = agg[(name, wgs)]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
with open(out, "w") as f:
    f.write(f"rocprofv3 --kernel-trace -- python tools/c5_step.py 10 {B}: e3dge kernels by (kernel, workgroups); avg us per launch, launches, share of the e3dge total\n")
    tot = sum(v[0] for _, v in rows)
    for (name, wgs), v in rows[:16]:
        f.write(f"{name[:70]:<70} {wgs:>6} wgs {v[0] / v[1] / 1e3:>9.1f} us {v[1]:>5} {100 * v[0] / tot:>6.2f} %\n")
print(open(out).read())
PY
  rm -rf "$OUT/b$B"
done
