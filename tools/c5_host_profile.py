"""cProfile of the stage-1 step's HOST side (tools/c5_step.py's step): where the Python time between launches goes."""
import cProfile, pstats, io, os, sys, runpy
sys.argv = [sys.argv[0], "3"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step.py"))
import torch
step = g["step"]
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
