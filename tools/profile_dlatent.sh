#!/bin/bash
# rocprofv3 kernel stats of the decoder forward + backward with d latent (packed path).  -> gpurun_out/dlatent_kernel_stats.txt
set -u
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dl" -o t -- python $REPO/tools/time_decoder_autograd.py --latent-grad --only-latent > "$OUT/dl.log" 2>&1
python - "$OUT/dl" "$OUT/dlatent_kernel_stats.txt" <<'PY'
import csv, glob, os, sys
d, out = sys.argv[1:3]
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(st[0])))
with open(out, "w") as f:
    for r in rows[:40]:
        f.write(f"{r['Name'][:100]:<100} {r['Calls']:>6} {float(r['AverageNs']):>10.0f} {r['Percentage']:>7}\n")
print(open(out).read())
PY
rm -rf "$OUT/dl"
tail -2 "$OUT/dl.log"
