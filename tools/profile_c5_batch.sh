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
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b$B" -o t -- python $REPO/tools/c5_step.py 10 $B > "$OUT/b$B.log" 2>&1
  python - "$OUT/b$B" "$OUT/kernel_stats_b$B.txt" <<'PY'
import csv, glob, os, sys
d, out = sys.argv[1:3]
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(st[0])))
with open(out, "w") as f:
    for r in rows[:14]:
        f.write(f"{r['Name'][:90]:<90} {r['Calls']:>6} {float(r['AverageNs']):>12.0f} {r['Percentage']:>7}\n")
print(open(out).read())
PY
  rm -rf "$OUT/b$B"
done
