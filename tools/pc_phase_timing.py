"""Per-phase cycle breakdown of the producer / consumer conv kernel (csrc/conv3x3_pc.hip) on single layers at bench size.

Measurement tool, not part of the product path: builds a second copy of the library with -DUNCL_PC_TIMING into tools/_timing/
(wave 0 = a consumer and wave 4 = a producer of every workgroup accumulate s_memtime deltas per loop phase) and prints the
share of each phase next to the launch time of both kernel structures.  Run on the GPU box:
    python tools/pc_phase_timing.py [--layers up3f,d0b,up1a,up0a,d1b,up2a]"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONS = ["MFMA phase", "epilogue", "barrier wait", "-"]
PROD = ["-", "register wait + LDS staging", "cursor + load issue", "barrier wait"]


def build():
    out = os.path.join(ROOT, "tools", "_timing")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libuncltmo_hip_pct.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "uncltmo_amd", "csrc", "*.hip")))
    deps = srcs + glob.glob(os.path.join(ROOT, "uncltmo_amd", "csrc", "*.h"))
    if os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(s) for s in deps):
        return lib
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(out, os.path.basename(s) + ".pct.o")
        objs.append(o)
        extra = os.environ.get("UNCL_PC_BUILD_FLAGS", "").split()
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DUNCL_PC_TIMING",
                                       *extra, "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


# name: (src_mode, skip/in channels, Cin, Cout, H (input extent seen by the conv), pad, pooled copy, coarse map)
LAYERS = {
    "up3f": ("up", 32, 128, 32, 252, 2, False),
    "up3": ("ssr", 32, 128, 32, 252, 2, False),
    "up2a": ("ssr", 64, 256, 32, 122, 2, False),
    "up1a": ("ssr", 128, 512, 64, 57, 2, False),
    "up0a": ("ssr", 256, 1024, 128, 24, 2, False),
    "d0a": ("plain", 32, 32, 64, 126, 0, False),
    "d0b": ("plain", 64, 64, 64, 124, 0, True),
    "d1a": ("plain", 64, 64, 128, 61, 0, False),
    "d1b": ("plain", 128, 128, 128, 59, 0, True),
    "d2a": ("plain", 128, 128, 256, 28, 0, False),
    "d2b": ("plain", 256, 256, 256, 26, 0, True),
    "up0b": ("plain", 128, 128, 128, 26, 2, False),
    "up3b": ("plain", 32, 32, 32, 254, 2, False),      # single chunk: UNCL_PC_NK1=1
    "inc1": ("plain", 32, 32, 32, 254, 0, False),
    "tail": ("tail", 32, 128, 32, 252, 2, False),       # the fused last decoder stage (up3f + second 3x3 + outconv in one launch)
    "o1c": ("o1c", 32, 32, 32, 254, 2, False),         # inference's last layer: plain source, only the fused 1x1 tail is stored (O1C)
    "incf": ("image1", 1, 32, 32, 256, 0, True),        # the product's first layer: inc.conv.conv rebuilt by the staging waves
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--layers", default="up3f,up2a,up1a,up0a,d0b,d1b,d2b")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--product", action="store_true", help="time the in-tree library (no phase stamps): for rocprofv3 --pmc runs")
    ap.add_argument("--only-pc", type=int, default=-1, help="with --product: run only this kernel structure (0 / 2)")
    args = ap.parse_args()
    lib_path = os.path.join(ROOT, "uncltmo_amd", "libuncltmo_hip.so") if args.product else build()
    if args.build_only:
        return
    import torch
    from uncltmo_amd import _hip
    _hip.LIB_PATH = lib_path
    lib = _hip.lib()
    if not args.product:
        lib.uncl_pc_timing_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        lib.uncl_pc_timing_read.restype = C.c_int
    bf = torch.bfloat16
    n = args.n
    g = torch.Generator(device="cuda").manual_seed(1)

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, device="cuda", generator=g) * scale).to(bf)

    for name in args.layers.split(","):
        mode, c, cin, cout, h, pad, pool = LAYERS[name]
        keep = []
        d = _hip.ConvDesc()
        x0 = rnd(n, h, h, c); keep.append(x0)
        if mode == "image1":
            # fp32 image (n, 256, 256); the conv sees the 254 x 254 x 32 map of inc.conv.conv
            x0 = torch.rand(n, h, h, device="cuda", generator=g); pw = torch.rand(32, 1, 3, 3, device="cuda", generator=g) * 0.3
            pb = torch.zeros(32, device="cuda"); keep += [x0, pw, pb]
            d.pre_w, d.pre_b = pw.data_ptr(), pb.data_ptr()
            d.src_mode = _hip.SRC_IMAGE1
        elif mode in ("up", "tail"):
            x1 = rnd(n, h // 2, h // 2, c); uw = rnd(4, 32, 32, scale=0.1); ub = torch.zeros(32, device="cuda"); keep += [x1, uw, ub]
            d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h // 2, h // 2, c
            d.up_w, d.up_b = uw.data_ptr(), ub.data_ptr()
            d.src_mode = _hip.SRC_CONCAT_SSR_UP
        elif mode == "ssr":
            h1 = h - 1 if h in (57,) else h
            x1 = rnd(n, h1, h1, c); keep.append(x1)
            d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h1, h1, c
            d.src_mode = _hip.SRC_CONCAT_SSR
        else:
            d.src_mode = _hip.SRC_PLAIN
        hc = h - 2 if mode == "image1" else h            # the extent the 3x3 layer itself sees
        ho = hc + 2 * pad - 2
        w = rnd(9, cout, cin, scale=0.05); b = torch.zeros(cout, device="cuda"); out = torch.empty(n, ho, ho, cout, dtype=bf, device="cuda")
        pl = torch.empty(n, ho // 2, ho // 2, cout, dtype=bf, device="cuda") if pool else None
        keep += [w, b, out, pl]
        d.dtype, d.ksize, d.pad, d.N, d.H, d.W, d.Cin, d.Cout = _hip.BF16, 3, pad, n, hc, hc, cin, cout
        d.src0, d.src0_H, d.src0_W, d.src0_C = x0.data_ptr(), h, h, c
        d.weight, d.bias, d.act = w.data_ptr(), b.data_ptr(), _hip.ACT_RELU
        d.out, d.out_H, d.out_W, d.out_C = out.data_ptr(), ho, ho, cout
        plp = pl.data_ptr() if pool else None
        if mode == "o1c":
            ow = torch.rand(32, device="cuda", generator=g) - 0.5; ob = torch.zeros(1, device="cuda"); o1 = torch.empty(n, ho, ho, device="cuda")
            keep += [ow, ob, o1]
            d.out1_w, d.out1_b, d.out1, d.out1_act = ow.data_ptr(), ob.data_ptr(), o1.data_ptr(), _hip.ACT_SIGMOID
            d.skip_main_store = 1
        if mode == "tail":
            w1 = rnd(9, 32, 32, scale=0.05); b1 = torch.zeros(32, device="cuda"); ow = torch.rand(32, device="cuda", generator=g) - 0.5
            ob = torch.zeros(1, device="cuda"); o1 = torch.empty(n, ho + 2, ho + 2, device="cuda"); keep += [w1, b1, ow, ob, o1]
            d.tail_w, d.tail_b, d.out1_w, d.out1_b, d.out1, d.out1_act = w1.data_ptr(), b1.data_ptr(), ow.data_ptr(), ob.data_ptr(), o1.data_ptr(), _hip.ACT_SIGMOID
            d.skip_main_store = 1
        gflop = 2.0 * 9 * cin * cout * ho * ho * n / 1e9

        def run(k):
            for _ in range(k):
                _hip.check(lib.uncl_conv3x3_pipe(C.byref(d), plp, _hip.stream_ptr()), "pipe")

        res = {}
        t = [0] * 32
        for pc in ((0, 2, 0, 2) if args.only_pc < 0 else (args.only_pc,)):
            lib.uncl_conv3x3_set_pc(pc)
            run(2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            buf = (C.c_ulonglong * 32)()
            if not args.product:
                lib.uncl_pc_timing_read(buf, 1)
            e0.record()
            run(args.reps)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(pc, []).append(e0.elapsed_time(e1) / args.reps)
            if pc == 2 and not args.product:
                lib.uncl_pc_timing_read(buf, 1)
                t = [int(buf[i]) for i in range(32)]
        lib.uncl_conv3x3_set_pc(2)
        r0, r2 = res.get(0, [float("nan")]), res.get(2, [float("nan")])
        print("== %s: four-wave %s ms, producer/consumer %s ms (%.0f / %.0f TFLOP/s)" % (
            name, ["%.3f" % v for v in r0], ["%.3f" % v for v in r2], gflop / min(r0), gflop / min(r2)))
        cons_names = ["MFMA of the four chunks", "image + result epilogues", "barrier wait", "second layer's MFMA"] if name == "tail" else CONS
        prod_names = ["the two idle barriers per tile"] + PROD[1:] if name == "tail" else PROD
        for label, names, base, cnt in (("consumer wave 0", cons_names, 0, t[8]), ("producer wave 4", prod_names, 4, t[9])):
            tot = sum(t[base:base + 4])
            if tot == 0 or cnt == 0:
                print("   %s: no samples (layer did not take the producer/consumer path)" % label)
                continue
            print("   %s: %d cycles per workgroup-launch" % (label, tot // cnt) + "".join(
                "; %s %.1f %%" % (nm, 100.0 * t[base + i] / tot) for i, nm in enumerate(names) if nm != "-"))
        if t[8] and t[9]:
            # per chunk kind (chunk index & 3; concat layers: the consumers multiply [x1, x2, x2^2, sqrt] = 0..3 while the producers
            # of iteration Q stage chunk Q + 1 and request chunk Q + 2): cycles per workgroup-launch
            print("   per chunk kind   consumer busy %s wait %s | producer (iteration Q) busy %s wait %s" % (
                [v // t[8] for v in t[24:28]], [v // t[8] for v in t[28:32]], [v // t[9] for v in t[16:20]], [v // t[9] for v in t[20:24]]))


if __name__ == "__main__":
    main()
