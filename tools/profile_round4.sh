#!/bin/bash
# Round-4 profile collection on the GPU box (one gpurun call): rocprofv3 kernel traces of (a) the headline at the default bench
# settings, (b) one decoder pass (packed pipeline), (c) the stage-1 training step; PMC passes (separate runs, --kernel-trace only) for
# the decoder convolutions' issue counters and the training step's HBM bytes.  Summaries land in gpurun_out/prof_r4/ -> copy to profiles/.
set -u
OUT=$PWD/gpurun_out/prof_r4
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
summ() {   # $1 = trace dir, $2 = output file, $3 = header line
python - "$1" "$2" "$3" <<'PY'
import csv, glob, os, sys
d, out, hdr = sys.argv[1:4]
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
with open(out, "w") as f:
    f.write(hdr + "\n")
    if not st:
        f.write("no kernel_stats.csv\n"); sys.exit(0)
    rows = list(csv.DictReader(open(st[0])))
    f.write(f"{'kernel':<100} {'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'pct':>7}\n")
    for r in rows[:45]:
        f.write(f"{r['Name'][:100]:<100} {r['Calls']:>6} {r['TotalDurationNs']:>14} {float(r['AverageNs']):>12.0f} {r['Percentage']:>7}\n")
print(open(out).read()[:3000])
PY
}
pmc() {   # $1 = tag, $2 = counters, rest = command
  local tag=$1 ctr=$2; shift 2
  timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$tag" -o pmc -- "$@" > "$OUT/pmc_$tag.log" 2>&1
  echo "pmc $tag rc=$?"
}
pmcsum() {  # $1 = output file, rest = tags
python - "$OUT" "$@" <<'PY'
import collections, csv, glob, os, sys
out, dst, tags = sys.argv[1], sys.argv[2], sys.argv[3:]
with open(dst, "w") as f:
    for tag in tags:
        c = glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True)
        if not c:
            f.write(f"{tag}: no counter csv\n"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(c[0])):
            k = r.get('Kernel_Name', r.get('Kernel Name', '?'))[:90]
            a = agg[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
        f.write(f"== {tag}\n")
        for k, cs in sorted(agg.items()):
            if any(x in k for x in ("pkconv", "pk_", "siren", "composite", "film", "resblock", "bwd_reduce")):
                f.write(k + "\n")
                for cn, (tot, n) in sorted(cs.items()):
                    f.write(f"    {cn:<32} mean/dispatch = {tot / max(n, 1):.6g}   (n={n})\n")
print(open(dst).read()[:4000])
PY
}
PARTS=${PARTS:-abcdpt}
if [[ $PARTS == *a* ]]; then
echo "== (a) headline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_headline" -o t -- python $REPO/bench.py --headline-only > "$OUT/trace_headline.log" 2>&1
summ "$OUT/trace_headline" "$OUT/r4_headline_kernel_stats.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --headline-only   (default K = 200, W = 20)"
fi; if [[ $PARTS == *b* ]]; then
echo "== (b) decoder pass"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_decoder" -o t -- python $REPO/tools/decoder_bench.py > "$OUT/trace_decoder.log" 2>&1
summ "$OUT/trace_decoder" "$OUT/r4_decoder_pass_kernel_stats.txt" "rocprofv3 --kernel-trace --stats -- python tools/decoder_bench.py   (1024^2, channel multiplier 2, packed pipeline; 50 passes)"
fi; if [[ $PARTS == *c* ]]; then
echo "== (c) training step"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_train" -o t -- python $REPO/tools/c5_step.py > "$OUT/trace_train.log" 2>&1
summ "$OUT/trace_train" "$OUT/r4_train_step_kernel_stats.txt" "rocprofv3 --kernel-trace --stats -- python tools/c5_step.py"
fi; if [[ $PARTS == *d* ]]; then
echo "== (d) evaluated images (two render passes with the backbone hand-over, texture head, decoder, metrics)"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o t -- python $REPO/tools/time_c3.py 128 > "$OUT/trace_c3.log" 2>&1
summ "$OUT/trace_c3" "$OUT/r4_c3_images_kernel_stats.txt" "rocprofv3 --kernel-trace --stats -- python tools/time_c3.py 128   (6 timed loops of 128 evaluated images, some on 2-3 streams)"
fi; if [[ $PARTS == *p* ]]; then
echo "== PMC (decoder)"
pmc dec_issue "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" python $REPO/tools/decoder_bench.py
pmc dec_lds "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" python $REPO/tools/decoder_bench.py
pmc dec_fetch "FETCH_SIZE" python $REPO/tools/decoder_bench.py
pmc dec_write "WRITE_SIZE" python $REPO/tools/decoder_bench.py
pmcsum "$OUT/r4_decoder_pmc.txt" dec_issue dec_lds dec_fetch dec_write
fi; if [[ $PARTS == *t* ]]; then
pmc train_fetch "FETCH_SIZE" python $REPO/tools/c5_step.py
pmc train_write "WRITE_SIZE" python $REPO/tools/c5_step.py
pmcsum "$OUT/r4_train_step_pmc.txt" train_fetch train_write
fi
find "$OUT" -type f -size +2M -delete
du -sh "$OUT"
