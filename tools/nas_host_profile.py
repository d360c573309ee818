"""Diagnostic: where the host time of a NAS supernet training step goes (cProfile over 20 steps, C5 shape)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NAS_STEPS", "3")
import nas_step as S                      # builds the model, runs warm-up + a few steps
import torch
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(20):
    S.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in buf.getvalue().splitlines()[4:42]))
