"""Timeline of one inversion forward from a rocprofv3 kernel trace of tools/inversion_host_profile.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/inversion_host_profile.py 30
    python tools/inversion_timeline.py OUT/t_kernel_trace.csv
Start, End -> Start boundary to the previous launch (signed, NOT clamped), duration (us) and name of every launch of one forward; the span, the
sum of the durations, and the boundaries split into positive (GPU idle between launches) and negative / zero ones.

How to read the boundaries: rocprofv3 takes Start / End from the dispatch packet's completion signal.  When launch n+1 is already queued
while launch n runs (the normal case here: the host enqueues faster than the GPU drains) the command processor stamps its start at the
moment it hands the packet over, which is the predecessor's end to the tick -- a boundary of exactly 0 does NOT mean that no time was
lost between the two kernels: the ~1.2-1.9 us of dispatch / wave launch / drain that MI355X_MICROARCH.md lists per kernel boundary sit
INSIDE the durations (the tail of n and the head of n+1).  Only boundaries > 0 are host-side starvation.  (Round 4's version clamped with
max(.., 0) and called the sum "idle": with every boundary at 0 that figure was an artefact of the stamps, as the round-4 review noted.)"""
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
prev, pos, neg, zero, busy = t0, 0, 0, 0, 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = s - prev                                        # signed: < 0 = overlaps its predecessor, 0 = stamped at its end (queued), > 0 = idle
    pos += max(g, 0)
    neg += min(g, 0)
    zero += g == 0
    name = re.sub(r'void |e3dge::|\(.*', '', r['Kernel_Name'])[:70]
    print(f"{(s - t0) / 1000:8.1f} us  boundary {g / 1000:+6.2f}  {(e - s) / 1000:7.1f} us  {name}")
    prev = e
    busy += e - s
print(f"one forward: span {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us, sum of launch durations {busy / 1000:.1f} us, {b - a} launches; "
      f"End->Start boundaries: {pos / 1000:.2f} us positive (GPU idle, host-side), {neg / 1000:.2f} us negative (overlap), {zero} stamped exactly at "
      f"the predecessor's end (queued launches: the per-boundary dispatch cost is inside the durations)")
