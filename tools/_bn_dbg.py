"""debug: accuracy of the forward's stage outputs (HIP fp32 vs oracle fp64 vs oracle fp32), batch_norm training mode"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_bnorm as T
from conftest import synth_state
from oracle import generator as OG
from uncltmo_amd import state_spec, synth
NAMES = ["INC0","X0","D0A","X1","D1A","X2","D2A","X3","D3A","X4","GFC1","GMR","GGC","GX1","FH","GOUT","U0UP","U0A","U0","U1UP","U1A","U1","U2UP","U2A","U2","U3UP","U3A","UPX","KNN","X0P","X1P","X2P","X3P","GGCZ","FHZ"]
DIMS = [(254,254,32),(252,252,32),(124,124,64),(122,122,64),(59,59,128),(57,57,128),(26,26,256),(24,24,256),(10,10,256),(12,12,256),
        (1,144,256),(1,144,512),(1,144,512),(1,144,256),(1,144,256),(1,144,256),(24,24,256),(26,26,128),(28,28,128),(56,56,128),(59,59,64),(61,61,64),
        (122,122,64),(124,124,32),(126,126,32),(252,252,32),(254,254,32),(256,256,32),(1,144,9),(126,126,32),(61,61,64),(28,28,128),(12,12,256),(1,144,512),(1,144,256)]
def layout(n, es):
    off, o = {}, 0
    for nm, d in zip(NAMES, DIMS):
        e = 4 if nm == "KNN" else es
        per = d[0]*d[1]*d[2]*e
        off[nm] = o
        o += (per*n + 255) & ~255
    return off
for norm in ("none", "batch_norm"):
    from uncltmo_amd.generator import UNet
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, norm, "none", "relu", 1, "replicate", 2, 0, compute_dtype="fp32")
    synth.fill_state_dict(net, "g0")
    if norm == "batch_norm": synth.bnorm_state(net.state_dict())
    net = net.cuda().train(); net.forced_drop_keep = [[1.0, 0.0], [1.0, 0.0]]
    x = T.inputs()
    with torch.no_grad():
        out, up, knn, ws = net._run(x.cuda().reshape(-1, 256, 256).float().contiguous(), need_feat=True, keep_act=True, save_preact=True)
    torch.cuda.synchronize()
    off = layout(2, 4)
    def buf(nm):
        d = DIMS[NAMES.index(nm)]
        n_el = 2*d[0]*d[1]*d[2]
        return ws[off[nm]:off[nm]+n_el*4].view(torch.float32).reshape(2, d[0], d[1], d[2]).permute(0,3,1,2).cpu()
    keep = torch.tensor([[1.0, 0.0], [1.0, 0.0]])
    wants = {}
    for dt in (torch.float32, torch.float64):
        sd = synth_state(state_spec.generator_spec(unet_norm=norm), "g0")
        if norm == "batch_norm": synth.bnorm_state(sd)
        full = {k: (v.clone().to(dt) if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
        w = {}
        with torch.no_grad():
            OG.unet_image_forward(full, x.to(dt), unet_norm=norm, training=True, drop_keep=keep, want=w)
        wants[dt] = w
    print("==", norm)
    for nm, key in (("X0","inc"),("X1","down0"),("X2","down1"),("X3","down2"),("X4","down3"),("U0","up0"),("U1","up1"),("U2","up2"),("UPX","up3")):
        h = buf(nm); r64 = wants[torch.float64][key]; r32 = wants[torch.float32][key]
        print("  %-5s hip-vs-64 rel %.2e maxabs %.2e   torch32-vs-64 rel %.2e maxabs %.2e   |x|max %.2e" % (key, T.rel(h, r64), (h.double()-r64).abs().max().item(), T.rel(r32, r64), (r32.double()-r64).abs().max().item(), r64.abs().max().item()))
