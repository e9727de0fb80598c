"""Host-side cost of one GanTrainerImg step: the step on 2 frames (GPU work ~1 ms) is bounded below by Python + launch time."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import model_factory, synth
from uncltmo_amd.optim import Adam
from uncltmo_amd.trainer_img import GanTrainer
dev = torch.device("cuda")
G = model_factory.create_G_net2("unet", dev, False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none", "none", "relu", True, 1, 1, 0,
                                "replicate", 2, 0, compute_dtype="bf16")
D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
synth.fill_state_dict(G, "g0"); synth.fill_state_dict(D, "d0"); G.train()
optG, optD = Adam(G.parameters(), lr=1e-5, betas=(0.5, 0.999)), Adam(D.parameters(), lr=1.5e-5, betas=(0.5, 0.999))
opt = types.SimpleNamespace(device=dev, pyramid_weight_list=torch.tensor([1.0, 1.0, 1.0]), ssim_loss_factor=1.0, ssim_window_size=5,
                            struct_method="gamma_ssim", add_frame=0, final_shape_addition=0, loss_g_d_factor=0.1,
                            adv_weight_list=torch.tensor([0.2, 0.2, 0.2]))
tr = GanTrainer(opt, G, D, optG, optD, None, None)
if os.environ.get("UNCL_SYNC_DEBUG"):      # list the host synchronisation points of one step
    hdr = synth.smooth_hdr_frames(4, salt="h").reshape(2, 2, 1, 256, 256).to(dev)
    pos = synth.ldr_frames(4, salt="p").reshape(2, 2, 1, 256, 256).to(dev)
    neg = (synth.ldr_frames(4, salt="n") ** 2).reshape(2, 2, 1, 256, 256).to(dev)
    tr.train_D(hdr, pos, neg, 0); tr.train_G(hdr, hdr, pos, neg, 0)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    tr.train_D(hdr, pos, neg, 0); tr.train_G(hdr, hdr, pos, neg, 0)
    torch.cuda.set_sync_debug_mode("default")
    sys.exit(0)
for B in (1, 16):
    hdr = synth.smooth_hdr_frames(B * 2, salt="h").reshape(B, 2, 1, 256, 256).to(dev)
    pos = synth.ldr_frames(B * 2, salt="p").reshape(B, 2, 1, 256, 256).to(dev)
    neg = (synth.ldr_frames(B * 2, salt="n") ** 2).reshape(B, 2, 1, 256, 256).to(dev)
    for _ in range(3):
        tr.train_D(hdr, pos, neg, 0); tr.train_G(hdr, hdr, pos, neg, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_D(hdr, pos, neg, 0); tr.train_G(hdr, hdr, pos, neg, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("N=%d frames: enqueue %.2f ms/step, with final sync %.2f ms/step" % (B * 2, (t1 - t0) * 100, (t2 - t0) * 100))
