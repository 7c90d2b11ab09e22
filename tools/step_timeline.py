"""Timeline of one stage-1 step from a rocprofv3 kernel trace of tools/c5_step.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/c5_step.py 12
    python tools/step_timeline.py OUT/t_kernel_trace.csv [--all]
Start (us from the step's first launch), duration, queue and name of every launch of the library (`--all`: also the framework's
small kernels), the step's span and how many small kernels it holds.  Shows what the event time of the step cannot: which stream
waits for the host and where."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'siren16_kernel<0, true' in r['Kernel_Name']]
if len(idx) < 3:
    sys.exit("fewer than three steps in the trace")
a, b = idx[len(idx) - 2], idx[len(idx) - 1]
t0 = int(rows[a]['Start_Timestamp'])
small = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name']
    ours = 'e3dge' in name
    if not ours:
        small += 1
        m = re.search(r'(FillFunctor|pow_tensor|MeanOps|NormTwoOps|CUDAFunctor\w*|AUnaryFunctor|BUnaryFunctor|BinaryFunctor|copyBuffer|direct_copy)', name)
        name = m.group(0) if m else name
    if ours or "--all" in sys.argv:
        print(f"{(s - t0) / 1000:8.1f} us  {(e - s) / 1000:7.1f} us  q{r['Queue_Id']}  {name[:90]}")
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us, {b - a} launches, {small} of them framework kernels")
