"""Per-phase cycle breakdown of conv3x3_pipe_kernel on the dominant layer (up_path.3.conv.conv, 200 tiles).

Measurement tool, not part of the product path: builds a second copy of the library with -DUNCL_PIPE_TIMING into
tools/_timing/ (wave 0 of every workgroup accumulates s_memtime deltas per loop phase) and prints the share of each
phase.  Run on the GPU box:  python tools/pipe_phase_timing.py [--layer up3|up3f|inc1|inc1f|up2|down1|mid]
"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ["cursor+prefetch issue", "MFMA phase", "barrier(before epilogue)", "epilogue LDS transpose+barrier",
          "global stores issue", "end-of-step barrier", "vmcnt(0) wait", "LDS staging writes", "barrier(after staging)"]


def build(extra=()):
    out = os.path.join(ROOT, "tools", "_timing")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libuncltmo_hip_timing%s.so" % ("_fi" if extra else ""))
    srcs = sorted(glob.glob(os.path.join(ROOT, "uncltmo_amd", "csrc", "*.hip")))
    if os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(s) for s in srcs):
        return lib
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-DUNCL_PIPE_TIMING", *extra,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "uncltmo_amd", "csrc"), "-o", lib] + srcs
    subprocess.check_call(cmd)
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--layer", default="up3")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--force-interior", action="store_true",
                    help="timing experiment: treat every tile as interior (border values are wrong, time is what an all-"
                         "interior layout would cost); inputs are placed in the middle of a larger allocation")
    args = ap.parse_args()
    lib_path = build(("-DUNCL_FORCE_INTERIOR",) if args.force_interior else ())
    if args.build_only:
        build(("-DUNCL_FORCE_INTERIOR",))
        return
    import torch
    from uncltmo_amd import _hip
    _hip.LIB_PATH = lib_path
    lib = _hip.lib()
    lib.uncl_pipe_timing_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.uncl_pipe_timing_read.restype = C.c_int
    bf = torch.bfloat16
    n = args.n
    g = torch.Generator(device="cuda").manual_seed(1)
    keep = []

    def rnd(*shape, scale=1.0):
        n_el = 1
        for d_ in shape:
            n_el *= d_
        big = (torch.rand(n_el + (1 << 22), device="cuda", generator=g) * scale).to(bf)   # 4 MiB of slack on either side
        keep.append(big)
        return big[1 << 21:(1 << 21) + n_el].reshape(*shape)

    d = _hip.ConvDesc()
    if args.layer == "up3":      # concat-ssr transposed 3x3, 128 -> 32 at 252^2 -> 254^2
        h, c, cin, cout, pad, mode = 252, 32, 128, 32, 2, _hip.SRC_CONCAT_SSR
        x1 = rnd(n, h, h, c); keep.append(x1)
        d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h, h, c
    elif args.layer == "up3f":   # the same layer with up_path.3.up (k2 s2, 32 -> 32) recomputed in the loader
        h, c, cin, cout, pad, mode = 252, 32, 128, 32, 2, _hip.SRC_CONCAT_SSR_UP
        x1 = rnd(n, h // 2, h // 2, c); keep.append(x1)
        uw = rnd(4, 32, 32, scale=0.1); ub = torch.zeros(32, device="cuda"); keep += [uw, ub]
        d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h // 2, h // 2, c
        d.up_w, d.up_b = uw.data_ptr(), ub.data_ptr()
    elif args.layer == "inc1":   # plain valid 3x3, 32 -> 32 at 254^2 -> 252^2 (static weights)
        h, c, cin, cout, pad, mode = 254, 32, 32, 32, 0, _hip.SRC_PLAIN
    elif args.layer == "inc1f":  # the same layer with inc.conv.conv recomputed from the fp32 image in the loader
        h, c, cin, cout, pad, mode = 254, 32, 32, 32, 0, _hip.SRC_IMAGE1
    elif args.layer == "up2":    # concat-ssr transposed 3x3, 256 -> 64 at 122^2 -> 124^2 (NT=2)
        h, c, cin, cout, pad, mode = 122, 64, 256, 64, 2, _hip.SRC_CONCAT_SSR
        x1 = rnd(n, h, h, c); keep.append(x1)
        d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h, h, c
    elif args.layer == "down1":  # plain valid 3x3, 32 -> 64 at 126^2 -> 124^2 (NT=2)
        h, c, cin, cout, pad, mode = 126, 32, 32, 64, 0, _hip.SRC_PLAIN
    else:                        # plain valid 3x3, 128 -> 128 at 28^2
        h, c, cin, cout, pad, mode = 28, 128, 128, 128, 0, _hip.SRC_PLAIN
    if mode == _hip.SRC_IMAGE1:
        x = torch.rand(n, h + 2, h + 2, device="cuda", generator=g); keep.append(x)
        pw = (torch.rand(32, 9, device="cuda", generator=g) - 0.5); pb = torch.zeros(32, device="cuda"); keep += [pw, pb]
        d.pre_w, d.pre_b = pw.data_ptr(), pb.data_ptr()
    else:
        x = rnd(n, h, h, c); keep.append(x)
    w = rnd(9, cout, cin, scale=0.05); b = torch.zeros(cout, device="cuda")
    ho = h + 2 * pad - 2
    out = torch.empty(n, ho, ho, cout, dtype=bf, device="cuda")
    d.dtype, d.ksize, d.pad, d.src_mode, d.N, d.H, d.W, d.Cin, d.Cout = _hip.BF16, 3, pad, mode, n, h, h, cin, cout
    d.src0, d.src0_H, d.src0_W, d.src0_C = x.data_ptr(), h, h, c
    if mode == _hip.SRC_IMAGE1:
        d.src0_H, d.src0_W, d.src0_C = h + 2, h + 2, 1
    d.weight, d.bias, d.act = w.data_ptr(), b.data_ptr(), _hip.ACT_RELU
    d.out, d.out_H, d.out_W, d.out_C = out.data_ptr(), ho, ho, cout
    buf = (C.c_ulonglong * 16)()
    for rep in range(args.reps + 1):
        lib.uncl_pipe_timing_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(lib.uncl_conv3x3_pipe(C.byref(d), None, _hip.stream_ptr()), "uncl_conv3x3_pipe")
        e1.record()
        torch.cuda.synchronize()
        lib.uncl_pipe_timing_read(buf, 0)
    ms = e0.elapsed_time(e1)
    vals = [buf[i] for i in range(9)]
    tot = float(sum(vals)) or 1.0
    wgs = buf[15]
    flop = 2.0 * 9 * cin * cout * ho * ho * n
    print("layer %s  N=%d  %.3f ms  %.0f TFLOP/s  workgroups=%d  cycles/WG=%.0f" % (args.layer, n, ms, flop / ms / 1e9, wgs, tot / max(wgs, 1)))
    for name, v in zip(PHASES, vals):
        print("  %-34s %6.2f %%   %10.0f cycles/WG" % (name, 100.0 * v / tot, v / max(wgs, 1)))


if __name__ == "__main__":
    main()
