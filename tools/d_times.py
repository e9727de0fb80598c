"""SimpleDiscriminator forward + backward time on 32 frames (HIP events) and gradient abs-sums; run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import model_factory, synth
dev = torch.device("cuda")
D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
synth.fill_state_dict(D, "d0")
x = synth.ldr_frames(32, salt="dt").reshape(32, 1, 256, 256).to(dev)
def fb():
    D.zero_grad()
    o, f = D(x)
    (o.sum() + 3 * f.sum()).backward()
for _ in range(5): fb()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): fb()
e1.record(); torch.cuda.synchronize()
g = [p.grad.double().abs().sum().item() for p in D.parameters()]
print("D fwd+bwd N=32: %.1f us" % (e0.elapsed_time(e1) * 1000 / 30), ["%.6e" % v for v in g])
