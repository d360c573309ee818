"""C5 workload (NAS supernet step: 16 blocks / 32 units / batch 32, bf16) replayed from a HIP graph: the step is host-bound on
its ~250 launches, and nothing in the training path synchronises with the host, so the whole step (forward, loss, backward,
capturable Adam) can be captured once with torch.cuda.CUDAGraph and replayed.  Prints eager and graphed ms per step and checks
that both routes produce the same parameters."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilesuperresolution_amd.models import get_model
dev = torch.device("cuda", 0)
ns = argparse.Namespace(model_type="NAS_MODEL", image_mean=0.5, num_channels=3, scale=4, num_blocks=16,
                        num_residual_units=32, width_search=True, pretrained=False, hot_dtype="bf16")


def make():
    torch.manual_seed(0)
    m = get_model(ns).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    return m, opt


x = torch.rand(32, 3, 48, 48, device=dev)
hr = torch.rand(32, 3, 192, 192, device=dev)


def step(m, opt):
    opt.zero_grad(set_to_none=True)
    out, speed = m(x)
    loss = torch.nn.functional.l1_loss(out, hr) + 0.1 * speed.sum()
    loss.backward()
    opt.step()
    return loss


n = int(os.environ.get("NAS_STEPS", 20))
# eager
me, oe = make()
for _ in range(3):
    step(me, oe)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step(me, oe)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n
# graphed: warm up on a side stream, capture one step, replay
mg, og = make()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step(mg, og)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
og.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss_g = step(mg, og)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
graphed = (time.perf_counter() - t0) / n
# both models have now taken 3 + n eager steps / 3 + 1 + 3 + n graphed steps on the same data: compare at equal step counts
mc, oc = make()
for _ in range(3 + 1 + 3 + n):
    step(mc, oc)
torch.cuda.synchronize()
d = (mg.flat.detach() - mc.flat.detach()).abs().max().item() / mc.flat.detach().abs().max().item()
print(f"NAS step: eager {eager * 1e3:.3f} ms, graph replay {graphed * 1e3:.3f} ms; max relative parameter difference after {7 + n} steps {d:.2e}")
