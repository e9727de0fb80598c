"""debug: dL/dz of every conv output in front of a BatchNorm -- HIP fp32 gradient arena vs oracle autograd (fp64, fp32)"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import test_gpu_bnorm as T
from _bn_dbg_layout import NAMES, DIMS, layout
from conftest import synth_state
from oracle import generator as OG
from uncltmo_amd import state_spec, synth
BUFS = ["INC0","X0","D0A","X1","D1A","X2","D2A","X3","D3A","X4","U0A","U0","U1A","U1","U2A","U2","U3A","UPX"]
qs = [q for _, q in state_spec.batch_norm_layers()]
net = T._train_net("fp32")
x = T.inputs(); wy = 0.5 + synth.smooth_hdr_frames(2, salt="bwy")
y, up = net(x.cuda())
((y * wy.cuda()).sum() + 1e-3 * up.float().sum()).backward()
torch.cuda.synchronize()
gws = list(net.__dict__["_gws_cache"].values())[0]
off = layout(2, 4)
def gbuf(nm):
    d = DIMS[NAMES.index(nm)]
    n_el = 2*d[0]*d[1]*d[2]
    return gws[off[nm]:off[nm]+n_el*4].view(torch.float32).reshape(2, d[0], d[1], d[2]).permute(0,3,1,2).cpu().double()
keep = torch.tensor([[1.0, 0.0], [1.0, 0.0]])
recs = {}
orig = OG._norm
for dt in (torch.float32, torch.float64):
    rec = {}
    def spy(xx, unet_norm, sd=None, q=None, training=False, rec=rec):
        xx.retain_grad(); rec[q] = xx
        return orig(xx, unet_norm, sd, q, training)
    OG._norm = spy
    sd = synth.bnorm_state(synth_state(state_spec.generator_spec(unet_norm="batch_norm"), "g0"))
    full = {}
    for k, v in sd.items():
        if v.dtype != torch.float32: full[k] = v.clone()
        elif k.endswith("relative_pos") or "running_" in k: full[k] = v.clone().to(dt)
        else: full[k] = v.clone().to(dt).requires_grad_(True)
    yo, upo = OG.unet_image_forward(full, x.to(dt), unet_norm="batch_norm", training=True, drop_keep=keep)
    ((yo * wy.to(dt)).sum() + 1e-3 * upo.sum()).backward()
    recs[dt] = {q: t.grad.double() for q, t in rec.items()}
OG._norm = orig
for q, b in reversed(list(zip(qs, BUFS))):
    h = gbuf(b); r64 = recs[torch.float64][q]; r32 = recs[torch.float32][q]
    print("%-34s %-5s hip-vs-64 %.2e   torch32-vs-64 %.2e   |g| %.2e  mean/rms %.2e" % (q, b, T.rel(h, r64), T.rel(r32, r64), r64.norm().item(), (r64.mean(dim=(0,2,3)).abs().mean() / r64.pow(2).mean().sqrt()).item()))
