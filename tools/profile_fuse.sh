#!/bin/bash
# rocprofv3 kernel stats of Fuse_sft_MLP's forward + backward with trainable parameters (HIP path). -> gpurun_out/fuse_trainable_kernel_stats.txt
set -u
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fu" -o t -- python $REPO/tools/time_fuse_autograd.py --profile-trainable > "$OUT/fu.log" 2>&1
python - "$OUT/fu" "$OUT/fuse_trainable_kernel_stats.txt" <<'PY'
import csv, glob, os, sys
d, out = sys.argv[1:3]
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(st[0])))
with open(out, "w") as f:
    for r in rows[:26]:
        f.write(f"{r['Name'][:110]:<110} {r['Calls']:>6} {float(r['AverageNs']):>10.0f} {r['Percentage']:>7}\n")
print(open(out).read())
PY
rm -rf "$OUT/fu"
tail -1 "$OUT/fu.log"
