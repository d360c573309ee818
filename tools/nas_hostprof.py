"""Diagnostic: cProfile of the host side of the NAS supernet training step."""
import cProfile, pstats, io, os, sys
os.environ["NAS_STEPS"] = "1"
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nas_step.py"))
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    ns["step"]()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
