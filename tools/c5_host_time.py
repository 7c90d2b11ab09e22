"""Host time of the stage-1 step (tools/c5_step.py's step) against its GPU time: enqueue-only wall time per step (no synchronisation inside the
loop; one at the end) vs the event-timed step.  When the two are close the step is bound by Python, not by the kernels."""
import os, sys, time, runpy
sys.argv = [sys.argv[0], "3"] + sys.argv[1:]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step.py"))
import torch
step = g["step"]
for _ in range(20):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms per step; with the final drain {1e3 * (t2 - t0) / n:.3f} ms per step")
# and with the GPU idle at every step's start (host latency fully exposed)
ts = []
for _ in range(20):
    torch.cuda.synchronize(); a = time.perf_counter(); step(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    ts.append((b - a, c - a))
ts.sort()
print(f"single step from idle: host {1e3 * ts[len(ts) // 2][0]:.3f} ms, to completion {1e3 * sorted(t[1] for t in ts)[len(ts) // 2]:.3f} ms")
