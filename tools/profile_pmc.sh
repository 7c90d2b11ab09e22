#!/bin/bash
# Issue-level PMC counters of one kernel family on the GPU box (separate passes, kernel-trace only -- see the task notes).
#   tools/profile_pmc.sh TAG KERNEL_SUBSTRING -- <command...>      -> gpurun_out/pmc_TAG/summary.txt
set -u
TAG=$1; FILT=$2; shift 3
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  (cd $R && timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- "$@" > "$OUT/g$i.log" 2>&1); echo "pmc group $i rc=$?"
done
python - "$OUT" "$FILT" <<'PY'
import csv, glob, os, sys, collections
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(os.path.join(out, "g*"))):
    if not os.path.isdir(d):
        continue
    for c in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(c)):
            k = r.get('Kernel_Name', '?')
            if filt in k:
                a = agg[k[:90]][r['Counter_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
with open(os.path.join(out, "summary.txt"), "w") as f:
    for k, cs in agg.items():
        f.write(k + "\n")
        for cn, (tot, n) in sorted(cs.items()):
            f.write(f"    {cn:<32} mean/dispatch = {tot / max(n, 1):.6g}   (n={n})\n")
print(open(os.path.join(out, "summary.txt")).read())
PY
find "$OUT" -type f -size +2M -delete
