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
json.dump(res, open(os.path.join(out, "traffic_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -type f -size +2M -delete
