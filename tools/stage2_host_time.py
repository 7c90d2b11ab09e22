"""Is the stage-2 step bound by the host?  Same step as tools/stage2_step.py; prints the wall time per step with the queue kept full
(one synchronisation at the end) beside the host time the launches of one step take (time until step() returns, GPU far behind)."""
import os, sys, time, runpy
sys.argv = [sys.argv[0], "2"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage2_step.py"))
import torch
step = ns["step"]
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for _ in range(20):
    a = time.perf_counter(); step(); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
host.sort()
print(f"stage-2 step: wall {1e3 * (t2 - t0) / 20:.3f} ms per step; host time of a step's launches: median {1e3 * host[10]:.3f} ms, min {1e3 * host[0]:.3f} ms; "
      f"queue drained {1e3 * (t2 - t1):.3f} ms after the last launch")
