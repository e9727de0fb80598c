"""Per-phase cycle breakdown of the flat-tile conv kernel (csrc/conv3x3_flat.hip) on single layers at bench size.

Measurement tool, not part of the product path.  Build the stamped library first (here, no GPU needed):
    FILES=conv3x3_flat tools/ab_variants.sh fltim "-DUNCL_FL_TIMING"
then on the GPU box:   python tools/fl_phase_timing.py [--layers up1a,up0a,d1a,d2a,up0b,up1b] [--flat 3]
Wave 0 (multiplying) and wave 4 (staging) of every workgroup stamp s_memtime per loop phase; printed per chunk kind (index & 3)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name: (src_mode, skip/in channels, Cin, Cout, H (input extent seen by the conv), pad)
LAYERS = {
    "up1a": ("ssr", 128, 512, 64, 57, 2), "up0a": ("ssr", 256, 1024, 128, 24, 2),
    "d1a": ("plain", 64, 64, 128, 61, 0), "d2a": ("plain", 128, 128, 256, 28, 0),
    "up0b": ("plain", 128, 128, 128, 26, 2), "up1b": ("plain", 64, 64, 64, 59, 2),
    "d3a": ("plain", 256, 256, 256, 12, 0), "d3b": ("plain", 256, 256, 256, 10, 2), "d2b": ("plain", 256, 256, 256, 26, 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="up1a,up0a,d1a,d2a,up0b,up1b")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--flat", type=int, default=1)
    ap.add_argument("--lib", default=os.path.join(ROOT, "tools", "_ab", "libfltim.so"))
    args = ap.parse_args()
    import torch
    from uncltmo_amd import _hip
    _hip.LIB_PATH = args.lib
    lib = _hip.lib()
    lib.uncl_fl_timing_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.uncl_fl_timing_read.restype = C.c_int
    bf = torch.bfloat16
    n = args.n
    g = torch.Generator(device="cuda").manual_seed(1)

    def rnd(*shape, scale=1.0):
        return (torch.rand(*shape, device="cuda", generator=g) * scale).to(bf)

    for name in args.layers.split(","):
        mode, c, cin, cout, h, pad = LAYERS[name]
        keep = []
        d = _hip.ConvDesc()
        x0 = rnd(n, h, h, c); keep.append(x0)
        if mode == "ssr":
            h1 = h - 1 if h == 57 else h
            x1 = rnd(n, h1, h1, c); keep.append(x1)
            d.src1, d.src1_H, d.src1_W, d.src1_C = x1.data_ptr(), h1, h1, c
            d.src_mode = _hip.SRC_CONCAT_SSR
        else:
            d.src_mode = _hip.SRC_PLAIN
        ho = h + 2 * pad - 2
        w = rnd(9, cout, cin, scale=0.05); b = torch.zeros(cout, device="cuda"); out = torch.empty(n, ho, ho, cout, dtype=bf, device="cuda")
        keep += [w, b, out]
        d.dtype, d.ksize, d.pad, d.N, d.H, d.W, d.Cin, d.Cout = _hip.BF16, 3, pad, n, h, h, cin, cout
        d.src0, d.src0_H, d.src0_W, d.src0_C = x0.data_ptr(), h, h, c
        d.weight, d.bias, d.act = w.data_ptr(), b.data_ptr(), _hip.ACT_RELU
        d.out, d.out_H, d.out_W, d.out_C = out.data_ptr(), ho, ho, cout
        gflop = 2.0 * 9 * cin * cout * ho * ho * n / 1e9

        def run(k):
            for _ in range(k):
                _hip.check(lib.uncl_conv3x3_pipe(C.byref(d), None, _hip.stream_ptr()), "pipe")

        res = {}
        t = [0] * 48
        for fl in (0, args.flat, 0, args.flat):
            lib.uncl_conv3x3_set_flat(fl)
            run(2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            buf = (C.c_ulonglong * 48)()
            lib.uncl_fl_timing_read(buf, 1)
            e0.record()
            run(args.reps)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(fl, []).append(e0.elapsed_time(e1) / args.reps)
            if fl != 0:
                lib.uncl_fl_timing_read(buf, 1)
                t = [int(buf[i]) for i in range(48)]
        r0, r1 = res[0], res[args.flat]
        print("== %s: rectangular %s ms, flat(%d) %s ms (%.0f / %.0f TFLOP/s)" % (
            name, ["%.3f" % v for v in r0], args.flat, ["%.3f" % v for v in r1], gflop / min(r0), gflop / min(r1)))
        nc, npd = t[32], t[33]
        if not nc or not npd:
            print("   no samples (layer did not take the flat path)")
            continue
        tot = sum(t[0:11])
        print("   multiplying wave 0: %d cycles per workgroup-launch; MFMA by chunk kind %s  barrier wait by kind %s  epilogue %d  tile setup %d  first wait %d" % (
            tot // nc, [v // nc for v in t[0:4]], [v // nc for v in t[4:8]], t[8] // nc, t[9] // nc, t[10] // nc))
        print("      shares: MFMA %.1f %%, barrier %.1f %%, epilogue %.1f %%, setup %.1f %%" % (
            100.0 * sum(t[0:4]) / tot, 100.0 * sum(t[4:8]) / tot, 100.0 * t[8] / tot, 100.0 * t[9] / tot))
        ptot = sum(t[16:28])
        print("   staging wave 4: %d cycles per workgroup-launch; by iteration Q: wait-for-registers + LDS staging %s  cursor + requests %s  barrier wait %s" % (
            ptot // npd, [v // npd for v in t[16:20]], [v // npd for v in t[20:24]], [v // npd for v in t[24:28]]))
        print("      shares: staging %.1f %%, requests %.1f %%, barrier %.1f %%" % (
            100.0 * sum(t[16:20]) / ptot, 100.0 * sum(t[20:24]) / ptot, 100.0 * sum(t[24:28]) / ptot))


if __name__ == "__main__":
    main()
