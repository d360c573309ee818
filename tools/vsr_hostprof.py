"""Diagnostic: cProfile of the host side of the BasicVSR propagation step."""
import cProfile, pstats, io, os, sys
sys.argv = [sys.argv[0]]
os.environ["VSR_STEPS"] = "1"
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vsr_step.py"))
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ns["step"]()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
