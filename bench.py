#!/usr/bin/env python3
"""bench.py — HDR frames/s of the tone-mapping generator on MI355X.

One step = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
    8 x 1024^2 log-compressed HDR frames -> 25 overlapping 256^2 tiles each (200 tiles) -> generator forward
    (bf16 MFMA) -> cross-fade back to 8 x 1024^2          [BASELINE.json configs[1]]
`value` is whole-job frames/s (all ranks); with --gpus N every rank runs its own 8 frames (weak scaling, the
tiles are independent so there is no data-path collective) and the slowest rank's time is used.

Extra objects on the JSON line:
  roofline     dense-bf16 MFMA roofline of the dominant kernel (the implicit-GEMM conv of up_path.3.conv.conv with
               up_path.3.up recomputed in its loader, 26.3 % of the generator's FLOPs): algorithmic FLOPs per launch / mean launch duration measured
               with HIP events inside the timed steps (`exclusive` = the same kernel in a single-stream forward, a
               cross-check).
  cpu_baseline the CPU oracle (a port, not the reference's own code) timed on this box's host cores on a bounded
               sample of the same workload (generator forward over 256^2 tiles), reported in the same unit.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GFLOP_PER_TILE = 18.2858            # 2*MAC of the 27 conv layers, one 256^2 tile
DOM_LAYER = 24                      # packed-weight index of up_path.3.conv.conv
# 64516 px x 32 x 1152 x 2 (the 3x3 over the 128-channel concat) + 15876 px x 32 x 128 x 2 (up_path.3.up, computed inside
# the same launch since r1g; the halo pixels it recomputes are not counted)
DOM_GFLOP_PER_TILE = 4.6820 + 0.1300
DOM_GFLOP_PER_TILE_F32 = 4.6820     # fp32 parity mode: separate up-conv launch
PEAK_BF16_TFLOPS = 2500.0           # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3
FRAMES, H, W, TILES_PER_FRAME = 8, 1024, 1024, 25


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("UNCL_CHUNK", "0")))
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "train_video"],
                    help="infer: BASELINE configs[1] (default, the headline metric); train: configs[2], one full "
                         "GanTrainerImg step on 32 frames of 256x256; train_video: configs[3], one GanTrainer step on clips "
                         "of T=5 (4 crops of 256x256 per 512x512 clip) -- both reported for DESIGN.md, not the headline")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the untimed single-stream pass that measures the dominant kernel alone (profiling runs: "
                         "keeps every launch of the trace in the product configuration)")
    return ap.parse_args()


def _flush_c_stdio():
    """RCCL prints its version banner through C stdio, which (piped) would otherwise be flushed at exit, after the JSON line."""
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def pmc_traffic(dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected separately over this same command, gfx950 correction applied; profiles/*_pmc_traffic.json).  bench.py
    cannot run the counters itself, so the value is the recorded one; None when there is no record for this dtype."""
    if dtype != "bf16":
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        doc = json.load(open(files[-1]))
        dom = doc["dominant"]
        for k in doc["kernels"]:
            if k["grid_x"] == dom["grid_x"] and "pipe_kernel<1, 4, 4, 4" in k["kernel"]:
                return int(k["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def cpu_baseline(seconds):
    """Oracle generator forward on the host cores, fp32, bounded to ~`seconds` of work."""
    from oracle.state import generator_state
    from oracle.generator import unet_image_forward
    from uncltmo_amd import synth
    # one thread per physical core of one socket at most: oversubscribing the small convolutions with every SMT
    # thread of the box (256 on the MI355X hosts) is >50x slower than 64 threads
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("UNCL_CPU_THREADS", "64"))))
    sd = generator_state("g0")
    x = synth.smooth_hdr_frames(4, salt="cpu")
    with torch.no_grad():
        unet_image_forward(sd, x[:1])                      # warm-up
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            unet_image_forward(sd, x)
            done += x.shape[0]
        dt = time.perf_counter() - t0
    tiles_per_s = done / dt
    return {"value": tiles_per_s / TILES_PER_FRAME, "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "%d 256x256 tile forwards of the fp32 CPU oracle in %.1f s (%.2f tiles/s); "
                                      "one 1024x1024 frame = 25 tiles" % (done, dt, tiles_per_s)}


def train_bench(a, rank, world, dist):
    """configs[2]: full image-trainer step (train_D + train_G, all losses, Adam) on N = 32 frames per rank;
    configs[3] (--mode train_video): the video trainer on 2 clips x 4 crops x T=5 = 40 frames per rank."""
    import types
    from uncltmo_amd import model_factory, synth
    from uncltmo_amd.distributed import DistributedOptimizer
    from uncltmo_amd.optim import Adam
    video = a.mode == "train_video"
    if video:
        from uncltmo_amd.trainer_vid import GanTrainer
    else:
        from uncltmo_amd.trainer_img import GanTrainer
    dev = torch.device("cuda", torch.cuda.current_device())
    make_g = model_factory.create_G_net if video else model_factory.create_G_net2
    G = make_g("unet", dev, False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none", "none", "relu",
               True, 1, 1, 0, "replicate", 2, 0, compute_dtype="bf16")
    D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
    synth.fill_state_dict(G, "g0")
    synth.fill_state_dict(D, "d0")
    G.train()
    optG, optD = Adam(G.parameters(), lr=1e-5, betas=(0.5, 0.999)), Adam(D.parameters(), lr=1.5e-5, betas=(0.5, 0.999))
    if dist:
        optG, optD = DistributedOptimizer(optG), DistributedOptimizer(optD)
    opt = types.SimpleNamespace(device=dev, pyramid_weight_list=torch.tensor([1.0, 1.0, 1.0]), ssim_loss_factor=1.0,
                                ssim_window_size=5, struct_method="gamma_ssim", add_frame=0, final_shape_addition=0,
                                loss_g_d_factor=0.1, adv_weight_list=torch.tensor([0.2, 0.2, 0.2]))
    tr = GanTrainer(opt, G, D, optG, optD, None, None)
    B, T = (8, 5) if video else (16, 2)     # video: 2 clips of 512x512 -> 4 spatial 256x256 crops each
    hdr = synth.smooth_hdr_frames(B * T, salt="tr%d" % rank).reshape(B, T, 1, 256, 256).to(dev)
    pos = synth.ldr_frames(B * T, salt="trp%d" % rank).reshape(B, T, 1, 256, 256).to(dev)
    neg = (synth.ldr_frames(B * T, salt="trn%d" % rank) ** 2).reshape(B, T, 1, 256, 256).to(dev)

    def step():
        tr.train_D(hdr, pos, neg, 0)
        tr.train_G(hdr, hdr, pos, neg, 0)

    for _ in range(a.warmup):
        step()
    if dist:
        import torch.distributed as td
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        ms = dt / a.steps * 1e3
        n = B * T
        # per frame: 2 generator forwards + 1 (summed) backward = 2*18.286 + 36.572 GFLOP (the reference runs 2 backwards)
        tfl = n * (2 * GFLOP_PER_TILE + 2 * GFLOP_PER_TILE) / ms
        name = "GanTrainer (video, T=5)" if video else "GanTrainerImg"
        _flush_c_stdio()
        print(json.dumps({"metric": "HDR frames/sec (256x256 full %s step)" % name, "value": world * n * a.steps / dt,
                          "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": ("GanTrainer video step (train_D + train_G, backward through time, all losses, "
                                                  "Adam), 2 clips x 4 crops x 5 frames of 256x256 per GPU, epoch regime 0 "
                                                  "(BASELINE.json configs[3])") if video else
                                                 ("GanTrainerImg step (train_D + train_G, all losses, Adam), 32 frames of 256x256 "
                                                  "per GPU, epoch regime 0 (BASELINE.json configs[2])"),
                                     "parallelism": "data-parallel x%d, gradient all-reduce" % world},
                          "generator_mfma": {"achieved": tfl, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": tfl / PEAK_BF16_TFLOPS,
                                             "note": "2 fwd + 1 summed bwd of G per frame, whole step time"},
                          "errD": tr.errD.item(), "errG_d": tr.errG_d.item(), "errG_struct": tr.errG_struct.item()}), flush=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = world > 1 or os.environ.get("UNCL_FORCE_DIST") == "1"   # the override lets a 1-GPU box exercise the RCCL path
    torch.cuda.set_device(local_rank)
    if dist:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from uncltmo_amd import _hip, synth, tiler
    from uncltmo_amd.generator import UNet
    if os.environ.get("UNCL_STREAMS"):        # experiments: 1 = everything on the caller's stream
        _hip.check(_hip.lib().uncl_gen_set_streams(int(os.environ["UNCL_STREAMS"])), "uncl_gen_set_streams")
    if a.mode in ("train", "train_video"):
        train_bench(a, rank, world, dist)
        if dist:
            td.destroy_process_group()
        return

    lib = _hip.lib()
    assert lib.uncl_device_ok() == 1, "bench needs an MI355X (gfx950)"
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
               "replicate", 2, 0, compute_dtype=a.dtype, chunk=a.chunk)
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    # synthetic frames (seeded, heavy-tailed log-compressed radiance), resident in HBM before timing starts
    frames = synth.hdr_frames(FRAMES, H, W, salt="bench%d" % rank).cuda()

    def step():
        return tiler.test_big_size_image2(frames, net, 0, 0, 0)

    for _ in range(a.warmup):
        step()
    lib.uncl_prof_enable(DOM_LAYER, 4096)
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    dt = time.perf_counter() - t0
    # per-launch durations of the dominant kernel, recorded by HIP events on the launch stream during the steps
    buf = (ctypes.c_float * 4096)()
    nrec = lib.uncl_prof_read(buf, 4096)
    dom_ms = sum(buf[i] for i in range(nrec)) / max(nrec, 1)
    tiles_per_launch = FRAMES * TILES_PER_FRAME * a.steps / max(nrec, 1)
    # The timed steps run the product configuration: four parts on four streams up to the third decoder stage, then the last
    # stage (the dominant launch) for all 200 tiles on one stream.  The same kernel in a purely single-stream forward is
    # measured separately (untimed) as a cross-check of the live figure.
    excl_ms, excl_tiles = 0.0, 0.0
    if not a.no_exclusive:
        lib.uncl_gen_set_streams(1)
        step()
        torch.cuda.synchronize()
        lib.uncl_prof_read(buf, 4096)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        nx = lib.uncl_prof_read(buf, 4096)
        excl_ms = sum(buf[i] for i in range(nx)) / max(nx, 1)
        excl_tiles = FRAMES * TILES_PER_FRAME * 3 / max(nx, 1)
        lib.uncl_gen_set_streams(4)
    lib.uncl_prof_enable(-1, 0)
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(out).all()

    if rank == 0:
        peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
        ms = dt / a.steps * 1e3
        fps = world * FRAMES * a.steps / dt
        dom_gflop = DOM_GFLOP_PER_TILE if a.dtype == "bf16" else DOM_GFLOP_PER_TILE_F32
        dom_tflops = dom_gflop * tiles_per_launch / dom_ms if dom_ms > 0 else 0.0
        excl_tflops = dom_gflop * excl_tiles / excl_ms if excl_ms > 0 else 0.0
        fwd_tflops = GFLOP_PER_TILE * FRAMES * TILES_PER_FRAME / ms          # per GPU
        traffic, traffic_src = pmc_traffic(a.dtype)
        line = {
            "metric": "HDR frames/sec (1024x1024 generator forward, tiled)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "UNet generator forward, batch 8 x 1024x1024 synthetic HDR -> 200 overlap tiles "
                                   "of 256x256 per GPU, eval mode, random-init weights (BASELINE.json configs[1])",
                       "frames_per_step_per_gpu": FRAMES, "tiles_per_frame": TILES_PER_FRAME, "chunk": a.chunk,
                       "parallelism": "frame-parallel x%d, no collective" % world},
            "roofline": {"bound": "mfma", "achieved": dom_tflops, "peak": peak, "unit": "TFLOP/s",
                         "frac": dom_tflops / peak, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         # skip 252^2 + coarse map 126^2 in, 254^2 out, 32 bf16 channels each
                         "algorithmic_bytes": int(tiles_per_launch * (252 * 252 + 126 * 126 + 254 * 254) * 64),
                         "hbm_gbps": (traffic / dom_ms / 1e6 if traffic and dom_ms > 0 else None),
                         "kernel": ("conv3x3_pipe_kernel<1,4,4,4,false,false>" if a.dtype == "bf16" else "conv_igemm_kernel<float,3,8,1,1>")
                                   + " @ up_path.3.conv.conv",
                         "launches": nrec, "avg_launch_ms": dom_ms, "tiles_per_launch": tiles_per_launch,
                         "gflop_per_tile": dom_gflop,
                         "concurrency": "the network up to the third decoder stage runs as four parts on four streams; the "
                                        "last stage (this launch) covers all tiles on one stream, nothing else in flight",
                         "exclusive": {"achieved": excl_tflops, "frac": excl_tflops / peak, "avg_launch_ms": excl_ms,
                                       "tiles_per_launch": excl_tiles,
                                       "note": "same kernel, single stream, 3 untimed steps after the timed region"}},
            "forward_mfma": {"achieved": fwd_tflops, "peak": peak, "unit": "TFLOP/s", "frac": fwd_tflops / peak,
                             "gflop_per_tile": GFLOP_PER_TILE, "note": "whole step incl. tiler, per GPU"},
        }
        if world == 1 and not a.no_cpu:
            line["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
