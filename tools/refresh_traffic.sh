#!/bin/bash
# Re-measures the HBM traffic of the headline render kernel for bench.py's roofline.traffic and stamps it with the source digest
# of the kernel files (bench.py reports the figure as STALE when the digest no longer matches) and the git commit.
#   on the GPU box (last gpurun of a round):   bash tools/refresh_traffic.sh <git-hash>
# Separate rocprofv3 --pmc passes per counter (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), --kernel-trace only, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled by the consumer (gfx950 tallies 128-B requests at 64 B).
set -u
GIT=${1:-unknown}
OUT=$PWD/gpurun_out/traffic
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for mode in f16x3 f32; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    E3DGE_MFMA_MODE=$mode timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/${mode}_$ctr" -o pmc -- \
      python $REPO/bench.py --steps 20 --warmup 3 --headline-only > "$OUT/${mode}_$ctr.log" 2>&1
    echo "$mode $ctr rc=$?"
  done
done
# matrix-pipe occupancy and the achieved shader clock of the headline kernel: one more pass, SQ + GRBM counters only
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/f16x3_busy" -o pmc -- \
  python $REPO/bench.py --steps 20 --warmup 3 --headline-only > "$OUT/f16x3_busy.log" 2>&1
echo "f16x3 busy rc=$?"
# HBM traffic of one stage-1 step (all its kernels), 23 steps per run of tools/c5_step.py
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/train_$ctr" -o pmc -- python $REPO/tools/c5_step.py 20 > "$OUT/train_$ctr.log" 2>&1
  echo "train $ctr rc=$?"
done
cd "$REPO"
python - "$OUT" "$GIT" <<'PY'
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
out, git = sys.argv[1], sys.argv[2]
res = {"source": "tools/refresh_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean per 64x64x24 render launch, KB",
       "git": git, "kernel_source_digest": bench.kernel_source_digest()}
for mode in ("f16x3", "f32"):
    ent = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        c = glob.glob(os.path.join(out, f"{mode}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        if not c:
            continue
        tot = n = 0
        for r in csv.DictReader(open(c[0])):
            k = r.get("Kernel_Name", r.get("Kernel Name", ""))
            if r["Counter_Name"] == ctr and ("siren16_kernel<0" in k or "siren_kernel<0" in k):
                tot += float(r["Counter_Value"]); n += 1
        if n:
            ent[ctr + "_KB"] = tot / n
            ent[ctr + "_launches"] = n
    if len(ent) >= 4:
        res[mode] = ent
# headline kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1,024 SIMDs; GRBM_GUI_ACTIVE = shader cycles of the dispatch;
# the kernel-trace of the same run gives its duration
c = glob.glob(os.path.join(out, "f16x3_busy", "**", "*counter_collection.csv"), recursive=True)
if c and "f16x3" in res:
    agg = collections.defaultdict(lambda: [0.0, 0])
    dur = [0.0, 0]
    for r in csv.DictReader(open(c[0])):
        k = r.get("Kernel_Name", r.get("Kernel Name", ""))
        if "siren16_kernel<0" in k:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[1] += 1
    m = {k: v[0] / max(v[1], 1) for k, v in agg.items()}
    if m.get("GRBM_GUI_ACTIVE"):
        e = res["f16x3"]
        e["SQ_INSTS_MFMA"] = m.get("SQ_INSTS_MFMA")
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0          # the counter comes back summed over the 8 XCDs (decoder PMC: 1.19e6 for a 64.5 us dispatch)
        e["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
        if dur[1]:
            e["shader_clock_ghz"] = cyc / (dur[0] / dur[1])
            e["profiled_kernel_us"] = dur[0] / dur[1] / 1e3
ent = {}
split = collections.defaultdict(dict)          # kernel -> {counter: KB per step}
import re
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    c = glob.glob(os.path.join(out, f"train_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    if c:
        rows = [r for r in csv.DictReader(open(c[0])) if r["Counter_Name"] == ctr]
        tot = sum(float(r["Counter_Value"]) for r in rows)
        # steps in the run = launches of the saving render forward (one per step; tools/c5_step.py warms the clock for 0.4 s before its timed steps)
        kname = lambda r: r.get("Kernel_Name", r.get("Kernel Name", ""))
        n_steps = sum(1 for r in rows if "siren16_kernel<0, true" in kname(r) or "siren_kernel<0, 0, true" in kname(r))
        ent[ctr + "_KB"] = tot / max(n_steps, 1)
        ent[ctr + "_steps"] = n_steps
        per = collections.defaultdict(float)
        for r in rows:
            k = kname(r)
            name = re.sub(r"void |e3dge::|\(.*", "", k) if "e3dge::" in k else "(framework kernels)"
            wgs = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
            per[(name, wgs if "e3dge::" in k else 0)] += float(r["Counter_Value"])
        for key, v in per.items():
            split[key][ctr] = v / max(n_steps, 1)
if len(ent) == 4:
    ent["note"] = "all kernels of tools/c5_step.py 20 (stage-1 steps 64x64x18, counted by their saving render forwards), per step"
    res["train_step"] = ent
    with open(os.path.join(out, "train_step_pmc_by_kernel.txt"), "w") as f:
        f.write("HBM traffic of one stage-1 step by (kernel, workgroups): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/c5_step.py 20;\n"
                "MB per step; FETCH doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md)\n")
        rows_ = sorted(split.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))
        tf = tw = 0.0
        for (name, wgs), v in rows_:
            fm, wm = 2 * v.get("FETCH_SIZE", 0) / 1024, v.get("WRITE_SIZE", 0) / 1024
            tf += fm; tw += wm
            if fm + wm >= 1.0:
                f.write(f"{name[:64]:<64} {wgs:>5} wgs  read {fm:9.1f} MB  write {wm:9.1f} MB\n")
        f.write(f"{'total':<64} {'':>9}  read {tf:9.1f} MB  write {tw:9.1f} MB  = {(tf + tw) / 1024:.2f} GB per step\n")
    print(open(os.path.join(out, "train_step_pmc_by_kernel.txt")).read())
json.dump(res, open(os.path.join(out, "traffic_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -type f -size +2M -delete
