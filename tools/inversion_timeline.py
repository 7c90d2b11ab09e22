"""Timeline of one inversion forward from a rocprofv3 kernel trace of tools/inversion_host_profile.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/inversion_host_profile.py 30
    python tools/inversion_timeline.py OUT/t_kernel_trace.csv
Start, gap to the previous launch's end, duration (us) and name of every launch of one forward; the span, the sum of the durations and the
sum of the gaps (the GPU's idle time inside a forward)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'siren16_kernel<0, false, 1>' in r['Kernel_Name']]
if len(idx) < 4:
    sys.exit("fewer than four forwards in the trace")
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev, gaps, busy = t0, 0, 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = max(s - prev, 0)
    gaps += g
    name = re.sub(r'void |e3dge::|\(.*', '', r['Kernel_Name'])[:70]
    print(f"{(s - t0) / 1000:8.1f} us  gap {g / 1000:5.1f}  {(e - s) / 1000:7.1f} us  {name}")
    prev = max(prev, e)
    busy += e - s
print(f"one forward: span {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us, sum of launch durations {busy / 1000:.1f} us, "
      f"idle between launches {gaps / 1000:.1f} us, {b - a} launches")
