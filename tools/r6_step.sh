#!/bin/bash
# Round 6: the stage-1 renderer step in the backward modes (event-timed), per-(kernel, grid) averages of the default under rocprofv3
mkdir -p gpurun_out
REPO=$PWD
for m in f16x3_g2 f16x3; do
  echo "== E3DGE_BWD_MODE=$m"; E3DGE_BWD_MODE=$m python tools/c5_step.py 40 1 2>&1 | tail -1; E3DGE_BWD_MODE=$m python tools/c5_step.py 20 4 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r6_step_modes.txt
export TMPDIR=/tmp
for m in f16x3_g2 f16x3; do
 for B in 1 4; do
  (cd /tmp && E3DGE_BWD_MODE=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_$m$B -o t -- python $REPO/tools/c5_step.py 10 $B > /dev/null 2>&1)
  python - gpurun_out/prof_$m$B gpurun_out/r6_c5_by_launch_${m}_b$B.txt $B $m <<'PY'
import collections, csv, glob, os, re, sys
d, out, B, m = sys.argv[1:5]
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
    f.write(f"E3DGE_BWD_MODE={m} rocprofv3 --kernel-trace -- python tools/c5_step.py 10 {B}: e3dge kernels by (kernel, workgroups); avg us per launch, launches, share\n")
    tot = sum(v[0] for _, v in rows)
    for (name, wgs), v in rows[:14]:
        f.write(f"{name[:70]:<70} {wgs:>6} wgs {v[0] / v[1] / 1e3:>9.1f} us {v[1]:>5} {100 * v[0] / tot:>6.2f} %\n")
print(open(out).read())
PY
  rm -rf gpurun_out/prof_$m$B
 done
done
