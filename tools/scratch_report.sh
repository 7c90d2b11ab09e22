#!/bin/bash
# Lists every kernel of the build with its VGPR / AGPR / SGPR / SGPR-spill / scratch use (hipcc -Rpass-analysis=kernel-resource-usage, device pass only).
#   bash tools/scratch_report.sh [extra -D flags]      -> profiles-style table on stdout; kernels with scratch are marked
D=cvpr23-e3dge_amd/csrc
FL="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -Wno-unused-result --cuda-device-only -Rpass-analysis=kernel-resource-usage"
T=$(mktemp -d)
for f in $D/*.hip; do (hipcc $FL "$@" -c $f -o $T/$(basename $f).o 2> $T/$(basename $f).log) & done; wait
python3 - $T <<'PY'
import glob, os, re, subprocess, sys
rows = []
for log in sorted(glob.glob(os.path.join(sys.argv[1], "*.log"))):
    cur = None
    for line in open(log):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"file": os.path.basename(log)[:-4], "name": m.group(1)}
            rows.append(cur)
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'file':<16} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'s-spill':>7} {'scratch':>7}  kernel")
for r, n in zip(rows, names):
    print(f"{r['file']:<16} {r.get('vgpr', 0):>4} {r.get('agpr', 0):>4} {r.get('sgpr', 0):>4} {r.get('sspill', 0):>7} {r.get('scratch', 0):>7}{' <-- SCRATCH' if r.get('scratch', 0) else ''}  {n[:140]}")
print(f"{sum(1 for r in rows if r.get('scratch', 0))} of {len(rows)} kernels use scratch; {sum(1 for r in rows if r.get('sspill', 0))} spill SGPRs to VGPR lanes "
      f"(max {max((r.get('sspill', 0) for r in rows), default=0)})")
PY
rm -rf $T
