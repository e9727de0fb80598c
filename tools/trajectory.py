"""bf16 fast path against the fp32 parity mode over a TRAJECTORY of optimisation steps (GanTrainerImg.py:200-339 / GanTrainer.py:202-338:
train_D + train_G, all losses, Adam), from identical initial weights on an identical sequence of synthetic batches.

Per step the bf16 generator's gradients are 3 - 8 % off per tensor in the encoder (tests/test_gpu_configs.py); what matters for
training is whether those errors are noise that Adam averages out or a bias that steers the run.  Printed: the three loss curves
of both runs (every `--every` steps) with their largest relative difference, and per network level after the last step
  drift   = |p_bf16 - p_fp32| / |p_fp32 - p_init|     (how far apart the two runs ended, in units of the distance travelled)
  cosine  = <p_bf16 - p_init, p_fp32 - p_init> / (| | | |)
`--yardstick` adds a third run: the fp32 mode again from weights moved by at most ONE fp32 ulp -- how far two correct fp32 runs
drift apart by themselves (Adam's normalised update turns a low-signal gradient element's sign into a full step, so a trajectory is
sensitive to perturbations of any size); the bf16 run's drift is to be read against that, not against zero.
Run on the GPU box:   python tools/trajectory.py [--steps 300] [--video] [--frames 32] [--batches 8] [--lr 1e-5] [--yardstick]
(the fp32 mode takes ~3 s per N = 32 step: 300 steps are a quarter of an hour).  `run()` is what tests/test_gpu_trajectory.py gates."""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def level(k):
    if k == "gcn.pos_embed":
        return k
    parts = k.split(".")
    return ".".join(parts[:2]) if parts[0] in ("down_path", "up_path") else parts[0]


def make_trainer(video, dtype, lr, perturb=0):
    import torch
    from uncltmo_amd import model_factory, synth
    from uncltmo_amd.optim import Adam
    if video:
        from uncltmo_amd.trainer_vid import GanTrainer
    else:
        from uncltmo_amd.trainer_img import GanTrainer
    dev = torch.device("cuda")
    make_g = model_factory.create_G_net if video else model_factory.create_G_net2
    G = make_g("unet", dev, False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none", "none", "relu", True, 1, 1, 0,
               "replicate", 2, 0, compute_dtype=dtype)
    D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
    synth.fill_state_dict(G, "g0")
    synth.fill_state_dict(D, "d0")
    if perturb:
        gen = torch.Generator().manual_seed(perturb)
        with torch.no_grad():
            for k, p in G.named_parameters():
                p.mul_((1 + 6e-8 * (2 * torch.rand(p.shape, generator=gen) - 1)).to(p.device))
    G.train()
    G.drop_path_prob = 0.0              # (DropPath draws would differ between two trainers that share one generator of random numbers)
    opt = types.SimpleNamespace(device=dev, pyramid_weight_list=torch.tensor([1.0, 1.0, 1.0]), ssim_loss_factor=1.0,
                                ssim_window_size=5, struct_method="gamma_ssim", add_frame=0, final_shape_addition=0,
                                loss_g_d_factor=0.1, adv_weight_list=torch.tensor([0.2, 0.2, 0.2]))
    tr = GanTrainer(opt, G, D, Adam(G.parameters(), lr=lr, betas=(0.5, 0.999)), Adam(D.parameters(), lr=1.5 * lr, betas=(0.5, 0.999)),
                    None, None)
    return tr, G, D


def batches(video, frames, n_batches):
    """the synthetic loader: `n_batches` fixed batches, visited round-robin (an epoch of n_batches iterations)"""
    import torch
    from uncltmo_amd import synth
    out = []
    for b in range(n_batches):
        if video:
            from uncltmo_amd.frame_util import clip_to_crops
            nclip = max(1, frames // 20)
            v = synth.hdr_frames(nclip * 5, 512, 512, salt="trajv%d" % b).reshape(nclip, 5, 1, 512, 512)
            p = synth.ldr_frames(nclip * 5, 512, 512, salt="trajp%d" % b).reshape(nclip, 5, 1, 512, 512)
            hdr, pos = clip_to_crops(v.cuda()), clip_to_crops(p.cuda())
            neg = pos ** 2
        else:
            B, T = max(1, frames // 2), 2
            hdr = synth.smooth_hdr_frames(B * T, salt="trajh%d" % b).reshape(B, T, 1, 256, 256).cuda()
            pos = synth.ldr_frames(B * T, salt="trajp%d" % b).reshape(B, T, 1, 256, 256).cuda()
            neg = (synth.ldr_frames(B * T, salt="trajn%d" % b) ** 2).reshape(B, T, 1, 256, 256).cuda()
        out.append((hdr, pos, neg))
    return out


def run(video=False, steps=300, frames=32, n_batches=8, lr=1e-5, every=10, verbose=True, yardstick=False):
    import torch
    data = batches(video, frames, n_batches)
    curves = {}
    params = {}
    init = None
    for dtype in ("fp32", "bf16") + (("fp32'",) if yardstick else ()):
        tr, G, D = make_trainer(video, dtype.rstrip("'"), lr, perturb=1 if dtype.endswith("'") else 0)
        if init is None:
            init = {k: p.detach().double().cpu().clone() for k, p in G.named_parameters()}
        c = []
        for i in range(steps):
            hdr, pos, neg = data[i % n_batches]
            tr.train_D(hdr, pos, neg, 0)
            tr.train_G(hdr, hdr, pos, neg, 0)
            if i % every == every - 1 or i == 0:
                c.append((i + 1, float(tr.errD.detach()), float(tr.errG_d.detach()), float(tr.errG_struct.detach())))
        curves[dtype] = c
        params[dtype] = {k: p.detach().double().cpu().clone() for k, p in G.named_parameters()}
        del tr, G, D
        torch.cuda.empty_cache()
    # loss curves: largest relative difference per curve
    worst = [0.0, 0.0, 0.0]
    for a, b in zip(curves["fp32"], curves["bf16"]):
        for j in range(3):
            worst[j] = max(worst[j], abs(b[1 + j] - a[1 + j]) / max(abs(a[1 + j]), 1e-12))
        if verbose:
            print("step %4d   fp32: errD %.5f errG_d %.5f errG_struct %.5f   bf16: errD %.5f errG_d %.5f errG_struct %.5f" % (a + b[1:]), flush=True)
    # parameters: drift and cosine per level
    lv = {}
    for k in init:
        if k.endswith("relative_pos"):
            continue
        d32, d16 = params["fp32"][k] - init[k], params["bf16"][k] - init[k]
        e = lv.setdefault(level(k), [0.0, 0.0, 0.0, 0.0])
        e[0] += float((d16 - d32).pow(2).sum()); e[1] += float(d32.pow(2).sum()); e[2] += float((d16 * d32).sum()); e[3] += float(d16.pow(2).sum())
    table = {name: {"drift": (e[0] / max(e[1], 1e-300)) ** 0.5, "cosine": e[2] / max((e[1] * e[3]) ** 0.5, 1e-300),
                    "travelled_fp32": e[1] ** 0.5} for name, e in lv.items()}
    if yardstick:
        ys = {}
        for k in init:
            if k.endswith("relative_pos"):
                continue
            d32, dp = params["fp32"][k] - init[k], params["fp32'"][k] - init[k]
            e = ys.setdefault(level(k), [0.0, 0.0])
            e[0] += float((dp - d32).pow(2).sum()); e[1] += float(d32.pow(2).sum())
        for name, e in ys.items():
            table[name]["drift_fp32_one_ulp"] = (e[0] / max(e[1], 1e-300)) ** 0.5
    if verbose:
        print("largest relative difference of the loss curves (bf16 vs fp32 mode): errD %.3e  errG_d %.3e  errG_struct %.3e" % tuple(worst))
        for name, t in table.items():
            print("%-16s drift %.4f   cosine %.5f   |p_fp32 - p_init| %.4e%s" % (
                name, t["drift"], t["cosine"], t["travelled_fp32"],
                "   fp32 run from one-ulp-perturbed weights: drift %.4f" % t["drift_fp32_one_ulp"] if "drift_fp32_one_ulp" in t else ""))
    return {"loss_rel_diff": dict(zip(("errD", "errG_d", "errG_struct"), worst)), "levels": table, "curves": curves}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--video", action="store_true")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--frames", type=int, default=0, help="frames per step (image: 32 = BASELINE configs[2]; video: 40 = 2 clips x 4 crops x T 5)")
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--yardstick", action="store_true")
    a = ap.parse_args()
    run(a.video, a.steps or (100 if a.video else 300), a.frames or (40 if a.video else 32), a.batches, a.lr, a.every, True, a.yardstick)
