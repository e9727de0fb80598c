#!/usr/bin/env python3
"""bench.py — HDR frames/s of the tone-mapping generator on MI355X.

One step = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
    8 x 1024^2 log-compressed HDR frames -> 25 overlapping 256^2 tiles each (200 tiles) -> generator forward
    (bf16 MFMA) -> cross-fade back to 8 x 1024^2          [BASELINE.json configs[1]]
`value` is whole-job frames/s (all ranks); with --gpus N every rank runs its own 8 frames (weak scaling, the
tiles are independent so there is no data-path collective) and the slowest rank's time is used.

Ranks.  `python bench.py --gpus N` with no WORLD_SIZE in the environment is the LAUNCHER: before touching a GPU it starts N
children of itself (one per LOCAL_RANK, 127.0.0.1 rendezvous), waits for them and exits with the worst exit code; asking for
more GPUs than are visible is refused with a message and exit code 2.  Under `python -m torch.distributed.run ... bench.py
--gpus N` the ranks already exist and RANK / LOCAL_RANK / WORLD_SIZE come from the environment.  Rank 0 prints the ONE JSON
line; at N > 1 it carries `ranks_seen` (an all-gather of the rank numbers over RCCL) and the per-rank step times.

Extra objects on the JSON line:
  roofline     dense-bf16 MFMA roofline of the WHOLE forward (algorithmic 18.286 GFLOP per tile x tiles / step time); its
               `dominant_kernel` sub-object is the same for the single largest launch (the implicit-GEMM conv of
               up_path.3.conv.conv with up_path.3.up recomputed in its loader, 26.3 % of the generator's FLOPs): algorithmic
               FLOPs per launch / mean launch duration measured with HIP events on the launch stream inside the timed steps.
  cpu_baseline the CPU oracle (a port, not the reference's own code) timed on this box's host cores on a bounded
               sample of the same workload (generator forward over 256^2 tiles), reported in the same unit.
  train_step / train_video_step   BASELINE configs[2] / configs[3] (gradient all-reduce over RCCL at N > 1): ms per step,
               frames/s and the generator's MFMA fraction in both FLOP conventions.
  leg_failures (N = 1, default run) the default single-GPU line is produced by THREE children, one per workload, started
               before this process makes a GPU call (run_legs): a child that dies is run once more and every failed attempt is
               listed here with its exit code and last stderr lines.  Under a launcher (WORLD_SIZE set) every rank starts one
               child per TRAINING leg before it touches its GPU (run_rank_training_legs: the children of a leg form their own
               process group on a shifted MASTER_PORT), then measures the headline region itself: a training leg that crashes
               or hangs costs its timeout and an entry here, not the line.  With --no-train, in --mode train / train_video and
               with UNCL_BENCH_INPROC=1 everything runs in the one process.
"""
import argparse
import contextlib
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_TILE = 18.2858            # 2*MAC of the 27 conv layers, one 256^2 tile
DOM_LAYER = 24                      # packed-weight index of up_path.3.conv.conv
# 64516 px x 32 x 1152 x 2 (the 3x3 over the 128-channel concat) + 15876 px x 32 x 128 x 2 (up_path.3.up, computed inside
# the same launch; the halo pixels it recomputes are not counted)
DOM_GFLOP_PER_TILE = 4.6820 + 0.1300
DOM_GFLOP_PER_TILE_F32 = 4.6820     # fp32 parity mode: separate up-conv launch
PEAK_BF16_TFLOPS = 2500.0           # MI355X dense bf16 / fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3
# workloads: frames per step per GPU, frame size, overlap tiles per frame (reference tiler loop, model_save_util.py:417-481)
WORKLOADS = {
    "1024": dict(frames=8, H=1024, W=1024, tiles=25,
                 name="UNet generator forward, batch 8 x 1024x1024 synthetic HDR -> 200 overlap tiles of 256x256 per GPU, eval "
                      "mode, random-init weights (BASELINE.json configs[1])"),
    "4k": dict(frames=1, H=2160, W=3840, tiles=220,
               name="4K HDR inference: whole 2160x3840 frames per GPU -> 220 overlap tiles of 256x256 each, tiled UNet forward, "
                    "eval mode, random-init weights (BASELINE.json configs[4]; quote it with --dtype fp16)"),
}
# The 26 packed-weight layers in uncl_gen_layer_name order: (name, Cin, Cout, output extent, taps, transposed 3x3).
# Two FLOP figures per layer.  `gflop` is SURVEY.md section 2.3A / 8(d)'s ALGORITHMIC count, the convention of GFLOP_PER_TILE: 2 MAC over
# the OUTPUT pixels of a convolution and over the INPUT pixels of a stride-1 transposed 3x3 (what a MAC hook on nn.ConvTranspose2d
# counts: 4682.0 MFLOP for up_path.3.conv.conv); `gflop_executed` is what the implicit GEMM multiplies, 2 MAC over the output pixels of
# every layer (4756.6 for that one: the transposed layers run as full-padding convolutions).  Fractions are quoted on the first.
LAYER_TABLE = [("inc.conv.conv1", 32, 32, 252, 9, False), ("down_path.0.mpconv.1.conv", 32, 64, 124, 9, False),
               ("down_path.0.mpconv.1.conv1", 64, 64, 122, 9, False), ("down_path.1.mpconv.1.conv", 64, 128, 59, 9, False),
               ("down_path.1.mpconv.1.conv1", 128, 128, 57, 9, False), ("down_path.2.mpconv.1.conv", 128, 256, 26, 9, False),
               ("down_path.2.mpconv.1.conv1", 256, 256, 24, 9, False), ("down_path.3.mpconv.1.conv", 256, 256, 10, 9, False),
               ("down_path.3.mpconv.1.conv1", 256, 256, 12, 9, True), ("gcn.module.0.0.fc1.0", 256, 256, 12, 1, False),
               ("gcn.module.0.0.graph_conv.gconv.nn.0", 128, 512, 12, 1, False), ("gcn.module.0.0.fc2.0", 512, 256, 12, 1, False),
               ("gcn.module.0.1.fc1.0", 256, 256, 12, 1, False), ("gcn.module.0.1.fc2.0", 256, 256, 12, 1, False),
               ("up_path.0.up", 256, 256, 24, 1, False), ("up_path.0.conv.conv", 1024, 128, 26, 9, True),
               ("up_path.0.conv.conv1", 128, 128, 28, 9, True), ("up_path.1.up", 128, 128, 56, 1, False),
               ("up_path.1.conv.conv", 512, 64, 59, 9, True), ("up_path.1.conv.conv1", 64, 64, 61, 9, True),
               ("up_path.2.up", 64, 64, 122, 1, False), ("up_path.2.conv.conv", 256, 32, 124, 9, True),
               ("up_path.2.conv.conv1", 32, 32, 126, 9, True), ("up_path.3.up", 32, 32, 252, 1, False),
               ("up_path.3.conv.conv", 128, 32, 254, 9, True), ("up_path.3.conv.conv1", 32, 32, 256, 9, True)]
# the two layers without a packed weight of their own (computed inside a neighbouring launch on every path): name, MFLOP per tile
FUSED_LAYERS = [("inc.conv.conv", 2.0 * 9 * 1 * 32 * 254 * 254 / 1e6), ("outc.conv", 2.0 * 1 * 32 * 1 * 256 * 256 / 1e6)]


def layer_gflops(tiles):
    """[(name, gflop in the survey's convention, gflop executed)] for `tiles` tile forwards; the first column sums to
    GFLOP_PER_TILE x tiles (tests/test_bench_launcher.py)."""
    rows = []
    for name, cin, cout, ho, taps, transposed in LAYER_TABLE:
        hs = ho - 2 if transposed else ho
        rows.append((name, 2.0 * taps * cin * cout * hs * hs * tiles / 1e9, 2.0 * taps * cin * cout * ho * ho * tiles / 1e9))
    for name, mflop in FUSED_LAYERS:
        rows.append((name, mflop * tiles / 1e3, mflop * tiles / 1e3))
    return rows


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--workload", default="1024", choices=sorted(WORKLOADS),
                    help="infer mode: 1024 = BASELINE configs[1] (default, the headline metric); 4k = configs[4], whole "
                         "2160x3840 frames per GPU (frame-parallel, no collective), normally with --dtype fp16")
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (default: the workload's)")
    ap.add_argument("--no-layers", action="store_true", help="skip the untimed per-layer pass behind roofline.layers")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("UNCL_CHUNK", "0")))
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "train_video"],
                    help="infer: BASELINE configs[1] (default, the headline metric; its line also carries train_step / "
                         "train_video_step); train: configs[2] alone, one full GanTrainerImg step on 32 frames of 256x256 per "
                         "GPU; train_video: configs[3] alone, one GanTrainer step on clips of T=5 (4 crops of 256x256 per "
                         "512x512 clip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-train", action="store_true", help="infer mode: skip the train_step / train_video_step legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="infer mode: after the timed region, repeat the step back to back for about this long and report "
                         "`sustained` (0: skip)")
    ap.add_argument("--no-eager", action="store_true", help="training legs: skip the eager steps timed beside the replayed graph "
                                                            "(profiling runs: every launch of the trace is a replayed one)")
    ap.add_argument("--no-4k", action="store_true", help="infer mode: skip the workload_4k sub-object (configs[4], fp16)")
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the untimed single-stream pass that measures the dominant kernel alone (profiling runs: "
                         "keeps every launch of the trace in the product configuration)")
    ap.add_argument("--video-clips", type=int, default=2,
                    help="train_video: 512x512 clips of T=5 per GPU and step (4 crops each: 2 -> 8, 8 -> 32 tiles per launch)")
    ap.add_argument("--leg", default=None, choices=["forward", "train_step", "train_video_step", "train_video_step_b8"],
                    help="(set by bench.py itself) run ONE leg of the default single-GPU line in this process and print its JSON: "
                         "the default run starts each leg as a child of its own, see run_legs()")
    ap.add_argument("--stub", action="store_true",
                    help="CPU self-test of the launcher and the JSON contract: gloo backend, the step is a small host matmul, "
                         "no GPU and no HIP library are touched (tests/test_bench_launcher.py)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh children, started before this process has made any GPU call
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPUs a child of this process may use, WITHOUT touching the HIP / HSA runtime (the launcher parent must not open the
    device before it forks its ranks): KFD topology nodes with simd_count > 0 (CPU nodes have 0), cut down by the
    *_VISIBLE_DEVICES lists the runtime itself would honour."""
    import re
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for d in os.listdir(base):
            try:
                props = open(os.path.join(base, d, "properties")).read()
            except OSError:
                continue
            m = re.search(r"^simd_count\s+(\d+)", props, re.M)
            if m and int(m.group(1)) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch(a, argv):
    """Parent of `python bench.py --gpus N` (N > 1, no WORLD_SIZE): start one child per GPU, pass rank 0's stdout through."""
    if not a.stub:
        have = visible_gpu_count()      # sysfs: this process never initialises the GPU runtime
        if a.gpus > have:
            sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run fewer ranks "
                             "than asked for\n" % (a.gpus, have))
            return 2
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(a.gpus),
               LOCAL_WORLD_SIZE=str(a.gpus))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
    rc = 0
    deadline = time.time() + float(os.environ.get("UNCL_BENCH_TIMEOUT", "1500"))
    for p in procs:
        try:
            code = p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            code = 124
            for q in procs:            # our own children, by PID
                if q.poll() is None:
                    q.kill()
        rc = max(rc, abs(code))
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# default single-GPU run: one child process per leg
# ---------------------------------------------------------------------------------------------------------------------
def _leg_command(argv, leg):
    return [sys.executable, os.path.abspath(__file__)] + list(argv) + ["--leg", leg]


def _last_json(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def _run_leg_child(argv, leg, env, attempts, deadline, failures):
    """one leg in a child of its own: its last JSON line, or None after `attempts` failures (each appended to `failures`)"""
    for attempt in range(1, attempts + 1):
        try:
            p = subprocess.run(_leg_command(argv, leg), env=env(attempt) if callable(env) else env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=max(1.0, deadline() - time.time()))
            rc, out, err = p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
        except subprocess.TimeoutExpired as e:
            rc, out, err = 124, (e.stdout or b"").decode(errors="replace"), (e.stderr or b"").decode(errors="replace")
        sys.stderr.write(err)
        doc = _last_json(out)
        if rc == 0 and (doc is not None or int(os.environ.get("RANK", "0")) != 0):
            return doc if doc is not None else {}
        failures.append({"leg": leg, "attempt": attempt, "rc": rc,
                         "stderr_tail": [l for l in err.splitlines() if "amdgpu.ids" not in l][-3:],
                         "fault": classify_fault(err)})
    return None


def dump_memory_map(tag):
    """Before a training leg's timed steps: every segment of the caching allocator (address, size, pool) and its blocks (offset,
    size, state) go to a file.  A leg that later dies with `Memory access fault by GPU ... on address 0x...` is then classified
    by its parent (classify_fault): inside a live block = a kernel wrote where it should not inside memory that is in use; inside
    an inactive / freed block = stream-ordering or use-after-free; within 2 MiB past a segment = an out-of-bounds index; nowhere
    near = not the allocator's memory (runtime, code objects, RCCL).  The library itself allocates nothing (workspaces are torch
    tensors), so the allocator's table is the whole table."""
    import torch
    path = os.environ.get("UNCL_BENCH_MEMMAP_DIR", "/tmp") + "/uncl_memmap_%s_%d.json" % (tag, os.getpid())
    segs = []
    for sg in torch.cuda.memory_snapshot():
        off, blocks = 0, []
        for b in sg.get("blocks", []):
            blocks.append([off, b["size"], b.get("state", "?")])
            off += b["size"]
        segs.append({"address": sg["address"], "size": sg["total_size"], "pool": list(sg.get("segment_pool_id", (0, 0))),
                     "stream": sg.get("stream", 0), "blocks": blocks})
    try:
        with open(path, "w") as f:
            json.dump({"pid": os.getpid(), "tag": tag, "segments": segs}, f)
        sys.stderr.write("[bench] memory map of leg %s: %s (%d segments)\n" % (tag, path, len(segs)))
    except OSError:
        pass
    return path


def classify_fault(stderr_text):
    """parent side: the faulting address of a dead leg against the memory map that leg wrote before its timed steps"""
    import re
    m = re.search(r"Memory access fault by GPU.*?on address (0x[0-9a-fA-F]+)", stderr_text, re.S)
    if not m:
        return None
    addr = int(m.group(1), 16)
    info = {"fault_address": m.group(1)}
    mm = re.findall(r"\[bench\] memory map of leg (\S+): (\S+) \(", stderr_text)
    if not mm:
        info["where"] = "no memory map was written before the fault"
        return info
    try:
        doc = json.load(open(mm[-1][1]))
    except (OSError, ValueError):
        info["where"] = "memory map unreadable"
        return info
    best = None
    for sg in doc["segments"]:
        lo, hi = sg["address"], sg["address"] + sg["size"]
        # the fault report is page-granular (4 KiB): compare pages
        if lo <= addr < hi or (lo >> 12) == (addr >> 12):
            off = addr - lo
            for boff, bsize, state in sg["blocks"]:
                if boff <= off < boff + bsize:
                    info["where"] = "inside a torch segment (pool %s): block +%d (%d B) state %s" % (sg["pool"], boff, bsize, state)
                    return info
            info["where"] = "inside a torch segment, no block"
            return info
        if hi <= addr < hi + (2 << 20) and (best is None or addr - hi < best[0]):
            best = (addr - hi, sg)
        if lo - (2 << 20) <= addr < lo and (best is None or lo - addr < best[0]):
            best = (lo - addr, sg)
    info["where"] = ("%d B outside the nearest torch segment (size %d, pool %s): out-of-bounds index" % (best[0], best[1]["size"], best[1]["pool"])
                     if best else "in no torch segment and not within 2 MiB of one: not allocator memory")
    return info


def run_legs(argv, legs=("forward", "train_step", "train_video_step", "train_video_step_b8"), attempts=2):
    """The default line carries three workloads (forward = the headline, train_step, train_video_step).  Each runs in a child of
    its own, started before this process has made any GPU call (like launch()): the three do not share an allocator history, and
    a leg that dies (round 3 saw ONE GPU memory fault in ~50 whole-bench runs, in a training leg, never reproduced under the
    uncached allocator, poisoned free memory or 40 repeats -- DESIGN.md) costs that leg one retry instead of the whole line.
    Every failed attempt is reported in the line's `leg_failures`; nothing is hidden and nothing is measured twice.
    The forward leg -- the line itself -- runs FIRST: with four legs of up to two 420 s attempts each, training legs that hang could
    otherwise eat the whole UNCL_BENCH_TIMEOUT budget and leave the headline a one-second deadline."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    end = time.time() + float(os.environ.get("UNCL_BENCH_TIMEOUT", "1500"))
    got, failures = {}, []
    per_leg = float(os.environ.get("UNCL_BENCH_LEG_TIMEOUT", "420"))      # a hung leg must not eat the whole budget
    for leg in legs:
        leg_end = [0.0]

        def deadline():
            if leg_end[0] == 0.0 or leg_end[0] < time.time():       # (re)armed at the start of every attempt
                leg_end[0] = min(end, time.time() + per_leg)
            return leg_end[0]
        doc = _run_leg_child(argv, leg, env, attempts, deadline, failures)
        if doc is not None:
            got[leg] = doc
    line = got.get("forward")
    if line is None:
        sys.stderr.write("bench.py: the forward leg produced no line\n")
        return 1
    for leg in legs:
        if leg != "forward":
            line[leg] = got.get(leg, {"error": "leg failed %d times, see leg_failures" % attempts})
    line["leg_failures"] = failures
    line["legs"] = ("one process per leg (forward, train_step, train_video_step at 2 clips per GPU, train_video_step_b8 at 8), started "
                    "by bench.py before any GPU call")
    print(json.dumps(line), flush=True)
    return 0


def run_rank_training_legs(argv, legs=("train_step", "train_video_step")):
    """Ranks of a launcher (WORLD_SIZE set): BEFORE this rank process touches its GPU, every rank starts the same child per
    training leg; the children of one leg form a process group of their own (same RANK / LOCAL_RANK / WORLD_SIZE, MASTER_PORT
    shifted per leg) and rank 0's child prints the leg's numbers.  A training leg that crashes or hangs (the RCCL gradient
    exchange has only ever run here at world size 1) then costs its own timeout and an entry in `leg_failures`, not the headline
    line, which this process measures afterwards in the launcher's own process group.  One attempt per leg: a retry would need
    the ranks to agree that the first one failed."""
    base = int(os.environ.get("MASTER_PORT", "29500"))
    per_leg = float(os.environ.get("UNCL_BENCH_LEG_TIMEOUT", "240"))
    got, failures = {}, []
    for k, leg in enumerate(legs):
        # torchrun's workers are told to use the AGENT's store on MASTER_PORT as clients; the children rendezvous on a port of
        # their own, so rank 0's child must host that store itself
        env = dict(os.environ, MASTER_PORT=str(base + 17 + 3 * k), TORCHELASTIC_USE_AGENT_STORE="False")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        end = time.time() + per_leg
        doc = _run_leg_child(argv, leg, env, 1, lambda: end, failures)
        got[leg] = {"error": "leg failed, see leg_failures"} if doc is None else doc
    return got, failures


# ---------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------
def _flush_c_stdio():
    """RCCL prints its version banner through C stdio, which (piped) would otherwise be flushed at exit, after the JSON line."""
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def trace(msg):
    """UNCL_BENCH_TRACE=1: leg markers on stderr (which leg a crash belongs to); stdout stays the one JSON line"""
    if os.environ.get("UNCL_BENCH_TRACE"):
        print("[bench] " + msg, file=sys.stderr, flush=True)


def pmc_traffic(dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected separately over this same command, gfx950 correction applied; profiles/*_pmc_traffic.json).  bench.py
    cannot run the counters itself, so the value is the recorded one; None when there is no record for this dtype."""
    if dtype != "bf16":
        return None, None
    import glob
    # the newest record by NAME (r<round><letter>_...): modification times do not survive a checkout
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")),
                   key=lambda f: (len(os.path.basename(f).split("_")[0]), os.path.basename(f)))
    if not files:
        return None, None
    try:
        doc = json.load(open(files[-1]))
        if "dominant" in doc and "hbm_bytes_per_launch" in doc["dominant"]:
            return int(doc["dominant"]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
        dom = doc["dominant"]
        for k in doc["kernels"]:
            if k["grid_x"] == dom["grid_x"] and "pipe_kernel<1, 4, 4, 4" in k["kernel"]:
                return int(k["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def cpu_baseline(seconds, tiles_per_frame=25, frame_name="1024x1024"):
    """Oracle generator forward on the host cores, fp32, bounded to ~`seconds` of work."""
    import torch
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
    return {"value": tiles_per_s / tiles_per_frame, "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "%d 256x256 tile forwards of the fp32 CPU oracle in %.1f s (%.2f tiles/s); "
                                      "one %s frame = %d tiles" % (done, dt, tiles_per_s, frame_name, tiles_per_frame)}


class Ranks:
    """The process group of this run: barrier, max-over-ranks timing and the self-check of who took part."""

    def __init__(self, a):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.stub = a.stub
        self.dist = self.world > 1 or os.environ.get("UNCL_FORCE_DIST") == "1"   # the override lets a 1-GPU box exercise RCCL
        import torch
        self.torch = torch
        if not self.stub:
            if self.local_rank >= torch.cuda.device_count():
                raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPU(s) visible" % (self.local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(self.local_rank)
        if self.dist:
            import torch.distributed as td
            self.td = td
            if self.stub:
                td.init_process_group("gloo")
            else:
                # RCCL's own stream at high priority: it then never shares the compute stream's hardware queue
                try:
                    opts = td.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                    td.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank), pg_options=opts)
                except (AttributeError, TypeError):
                    td.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        self.dev = torch.device("cpu") if self.stub else torch.device("cuda", self.local_rank)

    def sync(self):
        if self.dist:
            self.td.barrier()
        if not self.stub:
            self.torch.cuda.synchronize()

    def timed(self, step, steps, warmup):
        """W untimed steps, then EXACTLY K steps between barrier + device sync on both sides; returns
        (max-over-ranks seconds, per-rank seconds list, last step's result)."""
        out = None
        for _ in range(warmup):
            out = step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.sync()
        dt = time.perf_counter() - t0
        per_rank = [dt]
        if self.dist:
            t = self.torch.tensor([dt], device=self.dev, dtype=self.torch.float64)
            allt = [self.torch.zeros_like(t) for _ in range(self.world)]
            self.td.all_gather(allt, t)
            per_rank = [float(x.item()) for x in allt]
        return max(per_rank), per_rank, out

    def timed_steps(self, step, steps):
        """EXACTLY `steps` steps between barrier + device sync on both sides, like timed(), plus what one mean cannot show: an
        event pair around every step (device time of each step) and the host time each step took to enqueue.  Returns
        (max-over-ranks seconds, per-rank seconds, per-step device ms, per-step host-enqueue ms)."""
        torch = self.torch
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        host = []
        self.sync()
        t0 = time.perf_counter()
        for e0, e1 in evs:
            h0 = time.perf_counter()
            e0.record()
            step()
            e1.record()
            host.append((time.perf_counter() - h0) * 1e3)
        self.sync()
        dt = time.perf_counter() - t0
        per_rank = [dt]
        if self.dist:
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(self.world)]
            self.td.all_gather(allt, t)
            per_rank = [float(x.item()) for x in allt]
        return max(per_rank), per_rank, [e0.elapsed_time(e1) for e0, e1 in evs], host

    def ranks_seen(self):
        if not self.dist:
            return [self.rank]
        t = self.torch.tensor([self.rank], device=self.dev, dtype=self.torch.int64)
        allr = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.td.all_gather(allr, t)
        return sorted(int(x.item()) for x in allr)

    def close(self):
        if self.dist:
            # captured graphs that hold RCCL launches must be gone, and the device idle, before the communicator is torn down
            import gc
            gc.collect()
            if not self.stub:
                self.torch.cuda.synchronize()
            self.td.destroy_process_group()


def common_fields(a, rk, dt, per_rank):
    f = {"n_gpus": rk.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    if rk.dist:
        seen = rk.ranks_seen()
        f["ranks_seen"] = seen
        f["per_rank_ms_per_step"] = [t / a.steps * 1e3 for t in per_rank]
        assert seen == list(range(rk.world)), "ranks that took part %s != 0..%d" % (seen, rk.world - 1)
    return f


# ---------------------------------------------------------------------------------------------------------------------
# training steps (configs[2], configs[3])
# ---------------------------------------------------------------------------------------------------------------------
def make_trainer(rk, video, clips=2):
    import types
    import torch
    from uncltmo_amd import model_factory, synth
    from uncltmo_amd.distributed import DistributedOptimizer
    from uncltmo_amd.optim import Adam
    if video:
        from uncltmo_amd.trainer_vid import GanTrainer
    else:
        from uncltmo_amd.trainer_img import GanTrainer
    dev = rk.dev
    make_g = model_factory.create_G_net if video else model_factory.create_G_net2
    G = make_g("unet", dev, False, 1, "sigmoid", 32, "square_and_square_root", 4, 0, "none", "none", "relu",
               True, 1, 1, 0, "replicate", 2, 0, compute_dtype="bf16")
    D = model_factory.create_D_net(1, 16, dev, False, "none", True, "simpleD", 3, "none", 3, 0, 0, 0)
    synth.fill_state_dict(G, "g0")
    synth.fill_state_dict(D, "d0")
    G.train()
    optG, optD = Adam(G.parameters(), lr=1e-5, betas=(0.5, 0.999)), Adam(D.parameters(), lr=1.5e-5, betas=(0.5, 0.999))
    if rk.dist:
        optG, optD = DistributedOptimizer(optG, module=G), DistributedOptimizer(optD)
    opt = types.SimpleNamespace(device=dev, pyramid_weight_list=torch.tensor([1.0, 1.0, 1.0]), ssim_loss_factor=1.0,
                                ssim_window_size=5, struct_method="gamma_ssim", add_frame=0, final_shape_addition=0,
                                loss_g_d_factor=0.1, adv_weight_list=torch.tensor([0.2, 0.2, 0.2]))
    tr = GanTrainer(opt, G, D, optG, optD, None, None)
    if video:
        # configs[3]: 2 clips of T=5 at 512x512 per GPU -> four spatial 256x256 crops each = 8 clips of 256^2
        from uncltmo_amd.frame_util import clip_to_crops
        nclip = int(clips)
        vclips = synth.hdr_frames(nclip * 5, 512, 512, salt="trv%d" % rk.rank).reshape(nclip, 5, 1, 512, 512)
        pclips = synth.ldr_frames(nclip * 5, 512, 512, salt="trvp%d" % rk.rank).reshape(nclip, 5, 1, 512, 512)
        hdr = clip_to_crops(vclips.to(dev))
        pos = clip_to_crops(pclips.to(dev))
        neg = pos ** 2
        B, T = 4 * nclip, 5
    else:
        B, T = 16, 2
        hdr = synth.smooth_hdr_frames(B * T, salt="tr%d" % rk.rank).reshape(B, T, 1, 256, 256).to(dev)
        pos = synth.ldr_frames(B * T, salt="trp%d" % rk.rank).reshape(B, T, 1, 256, 256).to(dev)
        neg = (synth.ldr_frames(B * T, salt="trn%d" % rk.rank) ** 2).reshape(B, T, 1, 256, 256).to(dev)

    def step():
        tr.train_D(hdr, pos, neg, 0)
        tr.train_G(hdr, hdr, pos, neg, 0)

    tr._step_graph = None
    tr._step_graph_error = None
    # data-parallel steps are captured too (the gradient all-reduces then run in-stream at the end of the backward pass:
    # distributed.GradReducer.in_stream); UNCL_TRAIN_GRAPH=0 keeps the eager step with its overlapped exchange
    if os.environ.get("UNCL_TRAIN_GRAPH", "1") != "0":
        # one hipGraph replay per optimisation step (uncltmo_amd/step_graph.py); eager launches if the capture fails
        from uncltmo_amd.step_graph import StepGraph
        try:
            sg = StepGraph(tr, hdr, hdr, pos, neg, 0)
            step = sg.replay
        except Exception as e:          # noqa: BLE001 -- report and fall back: the eager step is the same arithmetic
            tr._step_graph = None
            tr._step_graph_error = "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()

    tr._eager_step = None if step is not None and getattr(tr, "_step_graph", None) is None else (
        lambda: (tr.train_D(hdr, pos, neg, 0), tr.train_G(hdr, hdr, pos, neg, 0)))
    return tr, step, B * T


def _median(v):
    v = sorted(v)
    n = len(v)
    return 0.0 if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


# the generator's seventeen 3x3 layers as the weight-gradient entry sees them: name, member channels (concat) or Cin, Cout,
# input H (= W), pad, concat
WGRAD3_LAYERS = [("inc.conv.conv1", 32, 32, 254, 0, 0), ("down_path.0.conv", 32, 64, 126, 0, 0), ("down_path.0.conv1", 64, 64, 124, 0, 0),
                 ("down_path.1.conv", 64, 128, 61, 0, 0), ("down_path.1.conv1", 128, 128, 59, 0, 0),
                 ("down_path.2.conv", 128, 256, 28, 0, 0), ("down_path.2.conv1", 256, 256, 26, 0, 0),
                 ("down_path.3.conv", 256, 256, 12, 0, 0), ("down_path.3.conv1", 256, 256, 10, 2, 0),
                 ("up_path.0.conv.conv", 256, 128, 24, 2, 1), ("up_path.0.conv.conv1", 128, 128, 26, 2, 0),
                 ("up_path.1.conv.conv", 128, 64, 57, 2, 1), ("up_path.1.conv.conv1", 64, 64, 59, 2, 0),
                 ("up_path.2.conv.conv", 64, 32, 122, 2, 1), ("up_path.2.conv.conv1", 32, 32, 124, 2, 0),
                 ("up_path.3.conv.conv", 32, 32, 252, 2, 1), ("up_path.3.conv.conv1", 32, 32, 254, 2, 0)]


def wgrad3_standalone(n, reps=5):
    """Weight + bias gradients of the generator's seventeen 3x3 layers at a batch of `n` frames, each launch timed ALONE with an
    event pair (uncl_conv_wgrad_bias on synthetic operands of the layer's shape, the dispatcher's default kernels): what the
    family costs a backward pass when nothing overlaps it.  The 1x1 / 2x2 layers' gradients (nine more launches) are not in it."""
    import ctypes as C
    import torch
    from uncltmo_amd import _hip
    lib = _hip.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    per, flops = {}, 0.0
    for name, c, cout, h, pad, cat in WGRAD3_LAYERS:
        cin = 4 * c if cat else c
        ho = h + 2 * pad - 2
        x = torch.rand(n, h, h, c, generator=g, device="cuda").to(torch.bfloat16)
        x1 = (torch.rand(n, h, h, c, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        gy = (torch.rand(n, ho, ho, cout, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        d = _hip.ConvDesc()
        kw = dict(dtype=_hip.BF16, ksize=3, pad=pad, src_mode=_hip.SRC_CONCAT_SSR if cat else _hip.SRC_PLAIN, N=n, H=h, W=h,
                  Cin=cin, Cout=cout, src0=x.data_ptr(), src0_H=h, src0_W=h, src0_C=c)
        if cat:
            kw.update(src1=x1.data_ptr(), src1_H=h, src1_W=h, src1_C=c)
        for k, v in kw.items():
            setattr(d, k, v)
        dw = torch.zeros(9, cout, cin, dtype=torch.float32, device="cuda")
        gb = torch.zeros(cout, dtype=torch.float32, device="cuda")
        call = lambda: _hip.check(lib.uncl_conv_wgrad_bias(C.byref(d), gy.data_ptr(), dw.data_ptr(), gb.data_ptr(), _hip.stream_ptr()),
                                  "uncl_conv_wgrad_bias")
        for _ in range(2):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        per[name] = round(e0.elapsed_time(e1) * 1e3 / reps, 1)
        flops += 2.0 * 9 * cin * cout * ho * ho * n
    tot_ms = sum(per.values()) * 1e-3
    return {"ms": tot_ms, "us_per_layer": per, "tflops": flops / (tot_ms * 1e-3) / 1e12, "frac_of_bf16_peak": flops / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
            "frames": n, "note": "the generator's seventeen 3x3 layers (weights + biases), each launch alone between HIP events, default "
                                 "kernels (split-role per pair, four-member on the skip-concat layers); 1x1 / 2x2 layers not included"}


def train_numbers(a, rk, video, steps, warmup, clips=None, leg=None):
    """One optimisation step (train_D + train_G) timed `steps` times after `warmup` untimed ones.  `ms_per_step` is the MEDIAN
    of the per-step device times; the mean over the bracketed region, min / max, the whole per-step list, the host
    time to enqueue a step and the garbage collector's activity are reported beside it, so that a number that moves between boxes
    (round 2: 8.4 ms here, 17.5 ms on the driver's box) can be told apart: a stall in one step, a host-bound step, or a slow GPU."""
    import gc
    import torch
    if os.environ.get("UNCL_POISON_GB"):          # debug: unwritten workspace memory reads as 0x7F bytes, not as zeros
        from uncltmo_amd.debug_poison import poison_free_memory
        poison_free_memory()
    trace("train leg (video=%s): building the trainer" % video)
    clips = clips or getattr(a, "video_clips", 2)
    tr, step, n = make_trainer(rk, video, clips)
    trace("trainer ready, graph=%s; warm-up" % (getattr(tr, "_step_graph", None) is not None))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    trace("timed steps")
    if not a.stub and os.environ.get("UNCL_BENCH_MEMMAP", "1") != "0":
        dump_memory_map(leg or ("train_video_step" if video else "train_step"))      # (a b8 fault is reported under its own leg's name)
    st0 = torch.cuda.memory_stats()
    gc0 = [g["collections"] for g in gc.get_stats()]
    dt, per_rank, dev_ms, host_ms = rk.timed_steps(step, steps)
    trace("timed steps done")
    gc1 = [g["collections"] for g in gc.get_stats()]
    st1 = torch.cuda.memory_stats()
    dev_allocs = st1.get("num_device_alloc", 0) - st0.get("num_device_alloc", 0)
    dev_frees = st1.get("num_device_free", 0) - st0.get("num_device_free", 0)
    ms_mean = dt / steps * 1e3
    ms = _median(dev_ms)
    # per frame: 2 generator forwards; backward = 2 x forward FLOPs per pass.  SURVEY §8(d) counts the reference's two
    # backward passes (109.7 GFLOP per frame); this build feeds the summed output gradient through ONE pass (73.1 GFLOP).
    tfl_1 = n * (2 * GFLOP_PER_TILE + 2 * GFLOP_PER_TILE) / ms
    tfl_2 = n * (2 * GFLOP_PER_TILE + 4 * GFLOP_PER_TILE) / ms
    out = {"ms_per_step": ms, "ms_median": ms, "ms_mean_wall": ms_mean, "ms_min": min(dev_ms), "ms_max": max(dev_ms),
           "ms_steps": [round(x, 3) for x in dev_ms],
           "host_enqueue_ms_median": _median(host_ms), "host_enqueue_ms_max": max(host_ms),
           "gc_collections_in_timed_steps": [b - c for b, c in zip(gc1, gc0)],
           "mode": "graph" if getattr(tr, "_step_graph", None) is not None else "eager",
           "graph_replay": bool(getattr(tr, "_step_graph", None) is not None),
           "graph_capture_error": getattr(tr, "_step_graph_error", None),
           "frames_per_s": rk.world * n / (ms * 1e-3), "frames_per_s_wall": rk.world * n * steps / dt,
           "frames_per_step_per_gpu": n, "steps": steps,
           "warmup": warmup, "dtype": "bf16", "device_mallocs_in_timed_steps": dev_allocs, "device_frees_in_timed_steps": dev_frees,
           "workload": ("GanTrainer video step (train_D + train_G, backward through time, all losses, Adam): %d clips of "
                        "512x512 x T=5 per GPU cut into 4 crops of 256x256 each = %d tiles per launch, epoch regime 0 "
                        "(BASELINE configs[3])" % (clips, 4 * clips)) if video
           else ("GanTrainerImg step (train_D + train_G, all losses, Adam): 32 frames of 256x256 per GPU, epoch regime 0 "
                 "(BASELINE configs[2])"),
           "generator_mfma": {"peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "achieved_executed": tfl_1, "frac_executed": tfl_1 / PEAK_BF16_TFLOPS,
                              "achieved_survey_convention": tfl_2, "frac_survey_convention": tfl_2 / PEAK_BF16_TFLOPS,
                              "note": "executed = 2 fwd + 1 summed bwd of G per frame (73.1 GFLOP); survey convention = 2 fwd + "
                                      "2 bwd (109.7 GFLOP, what the reference runs); both over the median step time"},
           "errD": float(tr.errD.detach()), "errG_d": float(tr.errG_d.detach()), "errG_struct": float(tr.errG_struct.detach())}
    if video:
        # the clip's frames in one workspace / arena, 3x3 / 2x2 weight gradients once per clip (uncltmo_amd/autograd.py)
        out["weight_gradients"] = ("once per clip (clip layout)" if os.environ.get("UNCL_CLIP_WGRAD", "1") != "0"
                                   else "per frame (UNCL_CLIP_WGRAD=0)")
    if not video and not a.stub and os.environ.get("UNCL_BENCH_WGRAD", "1") != "0":
        try:
            out["wgrad_ms"] = (w3 := wgrad3_standalone(n))["ms"]
            out["wgrad3x3_standalone"] = w3
        except Exception as e:        # a report beside the step, never the reason a leg fails
            out["wgrad3x3_standalone"] = {"error": repr(e)}
    if getattr(tr, "_eager_step", None) is not None and not getattr(a, "no_eager", False):
        # beside the replayed step: the same step launched eagerly (no hipGraph).  On an idle host it is the faster of the two on
        # the image step -- the generator's backward then runs its weight gradients on a second stream, which a replayed graph
        # cannot afford on this runtime -- on a slow host it is the host-bound one; `ms_per_step` stays the replay
        # (on the stream the graph was built on: the parameters' AccumulateGrad nodes belong to it, and autograd warns -- and
        # synchronises -- when a backward pass runs on another stream than the one they were created on)
        sg = getattr(tr, "_step_graph", None)
        ctx = torch.cuda.stream(sg.stream) if sg is not None else contextlib.nullcontext()
        if sg is not None:
            sg.stream.wait_stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(3):
                tr._eager_step()
            _, _, eager_ms, eager_host = rk.timed_steps(tr._eager_step, 10)
        if sg is not None:
            torch.cuda.current_stream().wait_stream(sg.stream)
        out["eager"] = {"ms_median": _median(eager_ms), "ms_min": min(eager_ms), "host_enqueue_ms_median": _median(eager_host),
                        "steps": 10}
    if rk.dist:
        # gradient exchange: bytes per step and the time the compute stream spent waiting in DistributedOptimizer.synchronize()
        # (collectives launched during the backward pass that had not finished when the optimiser asked for the gradients)
        from uncltmo_amd.distributed import exposed_allreduce_ms
        ex = exposed_allreduce_ms([tr.optimizerG, tr.optimizerD])
        ex = ex[-min(len(ex), 10 if out["mode"] == "graph" else steps):]      # graph mode: the eager comparison steps recorded these
        # the same bytes as one stand-alone all-reduce on the compute stream: what an in-stream exchange (graph mode) exposes
        nb = int(sum(p.numel() for p in tr.netG.parameters() if p.requires_grad))
        probe = torch.zeros(nb, dtype=torch.float32, device=rk.dev)
        pe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        rk.td.all_reduce(probe)
        for e0, e1 in pe:
            e0.record()
            rk.td.all_reduce(probe)
            e1.record()
        torch.cuda.synchronize()
        standalone = _median([e0.elapsed_time(e1) for e0, e1 in pe])
        out["allreduce"] = {"bytes_per_step": int(sum(p.numel() for p in tr.netG.parameters() if p.requires_grad) * 4 +
                                                   sum(p.numel() for p in tr.netD.parameters()) * 4),
                            "ms_exposed_median": _median(ex) if ex else None, "ms_exposed_max": max(ex) if ex else None,
                            "ms_exposed_is_of": "the eager steps (overlapped exchange on a side stream)",
                            "ms_standalone_generator_allreduce": standalone,
                            "exchange_in_timed_steps": "in-stream inside the replayed graph (exposed: about ms_standalone)"
                            if out["mode"] == "graph" else "overlapped with the backward pass on a side stream",
                            "world": rk.world}
    return out, per_rank, dt


def train_bench(a, rk):
    video = a.mode == "train_video"
    nums, per_rank, dt = train_numbers(a, rk, video, a.steps, a.warmup)
    if rk.rank == 0:
        name = "GanTrainer (video, T=5)" if video else "GanTrainerImg"
        # the line's value / ms_per_step follow the bench contract (K steps between two syncs, slowest rank); the per-step
        # distribution sits beside them
        line = {"metric": "HDR frames/sec (256x256 full %s step)" % name, "value": nums["frames_per_s_wall"], "unit": "frames/s"}
        line.update(common_fields(a, rk, dt, per_rank))
        line.update({"dtype": "bf16",
                     "config": {"workload": nums["workload"], "parallelism": "data-parallel x%d, gradient all-reduce" % rk.world},
                     "generator_mfma": nums["generator_mfma"],
                     "device_mallocs_in_timed_steps": nums["device_mallocs_in_timed_steps"],
                     "errD": nums["errD"], "errG_d": nums["errG_d"], "errG_struct": nums["errG_struct"]})
        for k in ("ms_median", "ms_mean_wall", "ms_min", "ms_max", "ms_steps", "host_enqueue_ms_median", "host_enqueue_ms_max",
                  "gc_collections_in_timed_steps", "mode", "graph_replay", "graph_capture_error", "allreduce", "eager"):
            if k in nums:
                line[k] = nums[k]
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
    elif rk.dist:
        rk.ranks_seen()


# ---------------------------------------------------------------------------------------------------------------------
# stub step (CPU tests of the launcher / JSON contract)
# ---------------------------------------------------------------------------------------------------------------------
def stub_bench(a, rk):
    import torch
    x = torch.ones(64, 64)

    def step():
        return (x @ x).sum()

    dt, per_rank, _ = rk.timed(step, a.steps, a.warmup)
    line = None
    if rk.rank == 0:
        line = {"metric": "stub steps/sec (launcher self-test, no GPU)", "value": rk.world * a.steps / dt, "unit": "steps/s"}
        line.update(common_fields(a, rk, dt, per_rank))
        line.update({"dtype": "f32", "config": {"workload": "stub", "parallelism": "x%d" % rk.world}})
        print(json.dumps(line), flush=True)
    elif rk.dist:
        rk.ranks_seen()


TRAIN_LINE_RANK_FIELDS = ("ranks_seen", "per_rank_ms_per_step", "mode", "allreduce")
ALLREDUCE_FIELDS = ("bytes_per_step", "ms_exposed_median", "ms_exposed_max", "ms_standalone_generator_allreduce",
                    "exchange_in_timed_steps", "world")


def stub_train_bench(a, rk):
    """--stub --mode train / train_video: the training line's multi-rank fields with a stand-in step (a small matmul and a gloo
    all-reduce of a 'gradient'): what the first N > 1 run on GPUs will print must be there -- ranks_seen, per-rank step times,
    `mode`, and the `allreduce` object (bytes, exposed time, stand-alone time, where the exchange ran).  CPU test of the contract
    only: no number of this line is a measurement of the product."""
    import torch
    w = torch.ones(256, 256)
    grad = torch.ones(1 << 16)
    exposed = []

    def step():
        y = (w @ w).sum()
        if rk.dist:
            t0 = time.perf_counter()
            rk.td.all_reduce(grad)
            exposed.append((time.perf_counter() - t0) * 1e3)
            grad.div_(rk.world)
        return y

    dt, per_rank, _ = rk.timed(step, a.steps, a.warmup)
    if rk.rank == 0:
        video = a.mode == "train_video"
        line = {"metric": "stub train steps/sec (launcher self-test, no GPU)", "value": rk.world * a.steps / dt, "unit": "steps/s"}
        line.update(common_fields(a, rk, dt, per_rank))
        line.update({"dtype": "f32", "mode": "eager", "graph_replay": False,
                     "config": {"workload": "stub %s step" % ("video" if video else "image"),
                                "parallelism": "data-parallel x%d, gradient all-reduce" % rk.world}})
        if rk.dist:
            ex = exposed[-a.steps:]
            line["allreduce"] = {"bytes_per_step": grad.numel() * 4, "ms_exposed_median": _median(ex), "ms_exposed_max": max(ex),
                                 "ms_exposed_is_of": "the stand-in step's in-line all-reduce",
                                 "ms_standalone_generator_allreduce": _median(ex),
                                 "exchange_in_timed_steps": "in line (stub)", "world": rk.world}
        print(json.dumps(line), flush=True)
    elif rk.dist:
        rk.ranks_seen()


# ---------------------------------------------------------------------------------------------------------------------
# headline: tiled generator forward
# ---------------------------------------------------------------------------------------------------------------------
def infer_bench(a, rk):
    import torch
    from uncltmo_amd import _hip, synth, tiler
    from uncltmo_amd.generator import UNet
    lib = _hip.lib()
    assert lib.uncl_device_ok() == 1, "bench needs an MI355X (gfx950)"
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
               "replicate", 2, 0, compute_dtype=a.dtype, chunk=a.chunk)
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    wl = WORKLOADS[a.workload]
    FRAMES, H, W, TILES_PER_FRAME = (a.frames or wl["frames"]), wl["H"], wl["W"], wl["tiles"]
    # synthetic frames (seeded, heavy-tailed log-compressed radiance), resident in HBM before timing starts
    frames = synth.hdr_frames(FRAMES, H, W, salt="bench%d" % rk.rank).cuda()

    def step():
        return tiler.test_big_size_image2(frames, net, 0, 0, 0)

    for _ in range(a.warmup):
        step()
    lib.uncl_prof_enable(DOM_LAYER, max(64, 2 * a.steps + 16))
    dt, per_rank, out = rk.timed(step, a.steps, 0)
    trace("forward timed")
    # per-launch durations of the dominant kernel, recorded by HIP events on the launch stream during the steps
    buf = (ctypes.c_float * 4096)()
    nrec = lib.uncl_prof_read(buf, 4096)
    dom_ms = sum(buf[i] for i in range(nrec)) / max(nrec, 1)
    tiles_per_launch = FRAMES * TILES_PER_FRAME * a.steps / max(nrec, 1)
    # Sustained figure: the timed region above is ~0.1 s, a burst for a chip whose clock follows its power draw; the same step
    # is repeated back to back for >= --sustain-seconds and every step is timed by an event pair on the stream.
    sustained = None
    if a.sustain_seconds > 0:
        est = dt / a.steps
        n_s = min(4000, max(200, int(a.sustain_seconds / est) + 1))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_s + 1)]
        rk.sync()
        evs[0].record()
        for i in range(n_s):
            step()
            evs[i + 1].record()
        rk.sync()
        s_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n_s)]
        k = min(100, n_s // 2)
        sustained = {"steps": n_s, "seconds": round(sum(s_ms) / 1e3, 3), "ms_per_step_median": _median(s_ms),
                     "ms_per_step_mean": sum(s_ms) / n_s, "first_100_ms": sum(s_ms[:k]) / k, "last_100_ms": sum(s_ms[-k:]) / k,
                     "ms_max": max(s_ms)}
        lib.uncl_prof_read(buf, 4096)             # drop the dominant-kernel records of these steps
        trace("sustained region done")
    # the fused last decoder stage (one launch, both 32-channel maps in LDS) beside the product default (two launches), same
    # process, interleaved: which of the two is the default was decided by this comparison
    tail_ab = None
    if a.dtype != "fp32" and not a.no_layers:
        res = {0: [], 1: []}
        old_tail = lib.uncl_gen_set_fused_tail(0)
        for rep in range(3):
            for mode in (0, 1):
                lib.uncl_gen_set_fused_tail(mode)
                step()
                rk.sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    step()
                rk.sync()
                res[mode].append((time.perf_counter() - t0) / 10 * 1e3)
        lib.uncl_gen_set_fused_tail(old_tail)
        lib.uncl_prof_read(buf, 4096)
        tail_ab = {"two_launches_ms_per_step": min(res[0]), "fused_ms_per_step": min(res[1]), "default": "fused" if old_tail else "two launches",
                   "note": "min of 3 x 10 steps each, interleaved; fused = up_path.3.conv.conv + conv1 + outc in one launch"}
    # The timed steps run the product configuration (several parts on several streams up to the third decoder stage, then the
    # last stage for all 200 tiles on one stream).  The same kernel in a purely single-stream forward is measured separately
    # (untimed) as a cross-check of the live figure.
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
        lib.uncl_gen_set_streams(int(os.environ.get("UNCL_STREAMS", "2")))
    # per-layer table (untimed, one stream, one layer's launches timed at a time): which layer is furthest below its roofline
    layers = None
    if not a.no_layers and a.dtype != "fp32":
        lib.uncl_gen_set_streams(1)
        layers = []
        peak_l = PEAK_BF16_TFLOPS
        table = layer_gflops(FRAMES * TILES_PER_FRAME)
        for i in range(len(LAYER_TABLE)):
            name, gfl, gfl_x = table[i]
            assert lib.uncl_gen_layer_name(i).decode() == name, (i, name)
            lib.uncl_prof_enable(i, 64)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            nl = lib.uncl_prof_read(buf, 4096)
            ent = {"layer": name, "gflop": round(gfl, 2), "gflop_executed": round(gfl_x, 2)}
            per_fwd = sum(buf[k] for k in range(nl)) / 2.0          # two forwards were timed
            if nl > 0 and per_fwd > 0:
                ent.update({"ms": round(per_fwd, 4), "tflops": round(gfl / per_fwd, 1), "frac": round(gfl / per_fwd / peak_l, 4),
                            "frac_executed": round(gfl_x / per_fwd / peak_l, 4)})
            else:
                ent.update({"ms": None, "note": "computed inside a neighbouring launch on this path"})
            layers.append(ent)
        for name, gfl, gfl_x in table[len(LAYER_TABLE):]:
            layers.append({"layer": name, "gflop": round(gfl, 2), "gflop_executed": round(gfl_x, 2), "ms": None,
                           "note": "computed inside a neighbouring launch on every path"})
        lib.uncl_gen_set_streams(int(os.environ.get("UNCL_STREAMS", "2")))
    lib.uncl_prof_enable(-1, 0)
    assert torch.isfinite(out).all()
    trace("exclusive / per-layer passes done")

    # BASELINE configs[4] beside the headline (driver-timed): one 2160x3840 frame per step = 220 tiles, fp16, the same tiler
    wl4k = None
    if a.workload == "1024" and a.dtype == "bf16" and not a.no_4k and rk.world == 1:
        w4 = WORKLOADS["4k"]
        net4 = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1,
                    "replicate", 2, 0, compute_dtype="fp16", chunk=a.chunk)
        synth.fill_state_dict(net4, "g0")
        net4 = net4.cuda().eval()
        fr4 = synth.hdr_frames(1, w4["H"], w4["W"], salt="bench4k").cuda()
        for _ in range(3):
            o4 = tiler.test_big_size_image2(fr4, net4, 0, 0, 0)
        n4 = 30
        e4 = [torch.cuda.Event(enable_timing=True) for _ in range(n4 + 1)]
        torch.cuda.synchronize()
        e4[0].record()
        for i in range(n4):
            o4 = tiler.test_big_size_image2(fr4, net4, 0, 0, 0)
            e4[i + 1].record()
        torch.cuda.synchronize()
        m4 = [e4[i].elapsed_time(e4[i + 1]) for i in range(n4)]
        assert torch.isfinite(o4).all()
        med4 = _median(m4)
        wl4k = {"workload": w4["name"], "dtype": "fp16", "tiles_per_frame": w4["tiles"], "steps": n4, "ms_per_frame_median": med4,
                "ms_per_frame_min": min(m4), "frames_per_s": 1e3 / med4,
                "roofline_frac": GFLOP_PER_TILE * w4["tiles"] / med4 / PEAK_BF16_TFLOPS,
                "note": "one frame per step, event pair per step, median of %d; `python bench.py --workload 4k --dtype fp16` is "
                        "the same workload as a line of its own" % n4}
        del net4, fr4, o4
        torch.cuda.empty_cache()
        trace("4k sub-workload done")

    train = {}
    if not a.no_train and a.dtype == "bf16":
        # the training legs are separate workloads: the inference model, its 6 GB workspace and the frames go first
        del out, step, net, frames
        torch.cuda.empty_cache()
        for key, video in (("train_step", False), ("train_video_step", True)):
            train[key], _, _ = train_numbers(a, rk, video, 30, 5)
            torch.cuda.empty_cache()
            trace("%s done" % key)

    if rk.rank != 0:
        if rk.dist:
            rk.ranks_seen()
        return
    peak = PEAK_F32_TFLOPS if a.dtype == "fp32" else PEAK_BF16_TFLOPS
    ms = dt / a.steps * 1e3
    fps = rk.world * FRAMES * a.steps / dt
    dom_gflop = DOM_GFLOP_PER_TILE_F32 if a.dtype == "fp32" else DOM_GFLOP_PER_TILE
    dom_tflops = dom_gflop * tiles_per_launch / dom_ms if dom_ms > 0 else 0.0
    excl_tflops = dom_gflop * excl_tiles / excl_ms if excl_ms > 0 else 0.0
    fwd_tflops = GFLOP_PER_TILE * FRAMES * TILES_PER_FRAME / ms          # per GPU
    # the roofline fraction is quoted on the SLOWER of the timed region and the sustained median (`value` stays the timed region)
    quoted_ms, quoted_on = ms, "timed region (%d steps)" % a.steps
    if sustained is not None and sustained["ms_per_step_median"] > ms:
        quoted_ms, quoted_on = sustained["ms_per_step_median"], "sustained median (%d steps, %.1f s)" % (sustained["steps"], sustained["seconds"])
    q_tflops = GFLOP_PER_TILE * FRAMES * TILES_PER_FRAME / quoted_ms
    traffic, traffic_src = pmc_traffic(a.dtype) if a.workload == "1024" else (None, None)
    mfma_name = {"bf16": "conv3x3 implicit-GEMM (bf16 MFMA)", "fp16": "conv3x3 implicit-GEMM (f16 MFMA)",
                 "fp32": "conv_igemm_kernel<float,3,8,1,1>"}[a.dtype]
    line = {"metric": "HDR frames/sec (%dx%d generator forward, tiled)" % (W, H) if a.workload != "1024" else
            "HDR frames/sec (1024x1024 generator forward, tiled)", "value": fps, "unit": "frames/s"}
    line.update(common_fields(a, rk, dt, per_rank))
    line.update({
        "dtype": a.dtype,
        "config": {"workload": wl["name"], "frame": "%dx%d" % (H, W),
                   "frames_per_step_per_gpu": FRAMES, "tiles_per_frame": TILES_PER_FRAME, "chunk": a.chunk,
                   "parallelism": "frame-parallel x%d, no collective" % rk.world},
        "roofline": {"bound": "mfma", "achieved": q_tflops, "peak": peak, "unit": "TFLOP/s", "frac": q_tflops / peak,
                     "quoted_on": quoted_on, "frac_timed_region": fwd_tflops / peak,
                     "scope": "whole forward: 18.2858 GFLOP per tile x %d tiles / ms per step (tiler included), per GPU" % (FRAMES * TILES_PER_FRAME),
                     "traffic": traffic, "traffic_unit": "bytes/launch of the dominant kernel",
                     "traffic_source": traffic_src,
                     "dominant_kernel": {
                         "kernel": mfma_name + " @ up_path.3.conv.conv (+ up_path.3.up in its loader)",
                         "achieved": dom_tflops, "frac": dom_tflops / peak,
                         # skip 252^2 + coarse map 126^2 in, 254^2 out, 32 bf16 channels each
                         "algorithmic_bytes": int(tiles_per_launch * (252 * 252 + 126 * 126 + 254 * 254) * 64),
                         "hbm_gbps": (traffic / dom_ms / 1e6 if traffic and dom_ms > 0 else None),
                         "launches": nrec, "avg_launch_ms": dom_ms, "tiles_per_launch": tiles_per_launch,
                         "gflop_per_tile": dom_gflop,
                         # (4.6820 GFLOP per tile for the 3x3 over the concatenation, + 0.1300 for the up-conv rebuilt in its
                         # loader; the implicit GEMM itself multiplies 4.7566 + the recomputed halo)
                         "frac_survey_convention": dom_tflops / peak,
                         "frac_survey_convention_conv_only": (DOM_GFLOP_PER_TILE_F32 * tiles_per_launch / dom_ms / peak if dom_ms > 0 else 0.0),
                         "exclusive": {"achieved": excl_tflops, "frac": excl_tflops / peak, "avg_launch_ms": excl_ms,
                                       "tiles_per_launch": excl_tiles,
                                       "note": "same kernel, single stream, 3 untimed steps after the timed region"}}},
    })
    if layers is not None:
        line["roofline"]["layers"] = layers
        line["roofline"]["layers_convention"] = ("gflop / tflops / frac: SURVEY 8(d)'s algorithmic count (transposed 3x3 layers over their INPUT "
                                                 "pixels; the column sums to 18.2858 GFLOP x tiles); gflop_executed / frac_executed: 2 MAC "
                                                 "over the output pixels, what the implicit GEMM multiplies")
    if sustained is not None:
        sustained["frac"] = GFLOP_PER_TILE * FRAMES * TILES_PER_FRAME / sustained["ms_per_step_median"] / peak
        line["sustained"] = sustained
    if wl4k is not None:
        line["workload_4k"] = wl4k
    if tail_ab is not None:
        line["fused_tail_ab"] = tail_ab
    line.update(train)
    if getattr(a, "rank_legs", None) is not None:
        line.update(a.rank_legs[0])
        line["leg_failures"] = a.rank_legs[1]
        line["legs"] = "training legs: one child per rank and leg with a process group of their own, before the headline region"
    if rk.world == 1 and not a.no_cpu:
        line["cpu_baseline"] = cpu_baseline(a.cpu_seconds, TILES_PER_FRAME, "%dx%d" % (H, W))
    _flush_c_stdio()
    print(json.dumps(line), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            return launch(a, argv)
        if (a.leg is None and not a.stub and a.mode == "infer" and not a.no_train and a.dtype == "bf16"
                and os.environ.get("UNCL_BENCH_INPROC") != "1"):
            return run_legs(argv)
    else:
        if int(os.environ["WORLD_SIZE"]) != a.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; the launcher's world size is used\n"
                             % (a.gpus, os.environ["WORLD_SIZE"]))
        if (a.leg is None and not a.stub and a.mode == "infer" and not a.no_train and a.dtype == "bf16"
                and os.environ.get("UNCL_BENCH_INPROC") != "1"):
            a.rank_legs = run_rank_training_legs(argv)      # children first: this process has not touched its GPU yet
            a.no_train = True
    rk = Ranks(a)
    try:
        if a.stub:
            if a.mode in ("train", "train_video"):
                stub_train_bench(a, rk)
            else:
                stub_bench(a, rk)
        else:
            from uncltmo_amd import _hip
            if os.environ.get("UNCL_STREAMS"):        # experiments: 1 = everything on the caller's stream
                _hip.check(_hip.lib().uncl_gen_set_streams(int(os.environ["UNCL_STREAMS"])), "uncl_gen_set_streams")
            if a.leg in ("train_step", "train_video_step", "train_video_step_b8"):
                nums, _, _ = train_numbers(a, rk, a.leg != "train_step", 30 if a.leg != "train_video_step_b8" else 12, 5 if a.leg != "train_video_step_b8" else 3,
                                           clips=8 if a.leg == "train_video_step_b8" else None, leg=a.leg)
                _flush_c_stdio()
                print(json.dumps(nums), flush=True)
            elif a.mode in ("train", "train_video"):
                train_bench(a, rk)
            else:
                if a.leg == "forward":
                    a.no_train = True
                infer_bench(a, rk)
    finally:
        rk.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
