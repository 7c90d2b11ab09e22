#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
python tools/stage2_step.py 10 2>&1 | tail -1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/s2 -o t -- python $R/tools/stage2_step.py 5 > /dev/null 2>&1)
python - <<'PY'
import collections, csv, glob, re
tr = glob.glob("/tmp/s2/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(tr[0])))
agg = collections.defaultdict(lambda: [0, 0])
steps = sum(1 for r in rows if "siren16_kernel<0, true" in r["Kernel_Name"])
for r in rows:
    name = re.sub(r"void |e3dge::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70] if not name.startswith("at::") else re.sub(r"at::native::|\(anonymous namespace\)::|std::array<char\*, \d+ul>|<unnamed>::", "", name)[:110]
    a = agg[name]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(v[0] for v in agg.values())
out = [f"stage-2 step (tools/stage2_step.py): kernel time per step by kernel name, {steps} steps in the trace; total {tot / steps / 1e6:.3f} ms of kernels per step"]
for name, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(__import__('os').environ.get('TOPN', '28'))]:
    out.append(f"{name:<70} {v[0] / steps / 1e3:>9.1f} us/step {v[1] / steps:>6.1f} launches/step {100 * v[0] / tot:>6.2f} %")
open("gpurun_out/r6_stage2_step_by_kernel.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
