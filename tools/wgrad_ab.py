"""Same-box A/B of the 3x3 weight-gradient kernels on the generator's own layer shapes at the image step's batch (N = 32):
per layer, microseconds per launch of the six-wave per-pair kernel (uncl_wgrad_set_roll(0), uncl_wgrad_set_cat(0)), the split-role
per-pair kernel (roll 2, cat 0, quad 0) and what the dispatcher picks by default (roll 1, cat 1, quad 1: the four-member kernel on the
skip-concat layers, 64 x 64 blocks on the plain layers that have them), plus the rel-L2 distance of the default result from the six-wave one.
   python tools/wgrad_ab.py [N]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from uncltmo_amd import _hip  # noqa: E402

import bench  # noqa: E402

LAYERS = bench.WGRAD3_LAYERS


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = _hip.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    tot = {0: 0.0, 1: 0.0, 2: 0.0}
    print("%-24s %9s %9s %9s  %s" % ("layer (us per launch)", "six-wave", "split", "default", "rel-L2 default vs six-wave"))
    for name, c, cout, h, pad, cat in LAYERS:
        cin = 4 * c if cat else c
        ho = h + 2 * pad - 2
        x = (torch.rand(n, h, h, c, generator=g, device="cuda") * (1 if cat else 2) - (0 if cat else 1)).to(torch.bfloat16)
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
        res, us = {}, {}
        for mode in (0, 2, 1):
            old = lib.uncl_wgrad_set_roll(mode)
            oldc = lib.uncl_wgrad_set_cat(1 if mode == 1 else 0)
            oldq = lib.uncl_wgrad_set_quad(1 if mode == 1 else 0)
            for rep in range(2):
                _hip.check(lib.uncl_conv_wgrad_bias(C.byref(d), gy.data_ptr(), dw.data_ptr(), gb.data_ptr(), _hip.stream_ptr()), "wgrad")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 6
            e0.record()
            for rep in range(reps):
                _hip.check(lib.uncl_conv_wgrad_bias(C.byref(d), gy.data_ptr(), dw.data_ptr(), gb.data_ptr(), _hip.stream_ptr()), "wgrad")
            e1.record()
            torch.cuda.synchronize()
            us[mode] = e0.elapsed_time(e1) * 1e3 / reps
            dw.zero_(); gb.zero_()
            _hip.check(lib.uncl_conv_wgrad_bias(C.byref(d), gy.data_ptr(), dw.data_ptr(), gb.data_ptr(), _hip.stream_ptr()), "wgrad")
            torch.cuda.synchronize()
            res[mode] = (dw.clone(), gb.clone())
            dw.zero_(); gb.zero_()
            lib.uncl_wgrad_set_roll(old)
            lib.uncl_wgrad_set_cat(oldc)
            lib.uncl_wgrad_set_quad(oldq)
            tot[mode] += us[mode]
        rel = float((res[1][0] - res[0][0]).norm() / res[0][0].norm())
        relb = float((res[1][1] - res[0][1]).norm() / res[0][1].norm())
        print("%-24s %9.1f %9.1f %9.1f  dw %.2e  gb %.2e" % (name, us[0], us[2], us[1], rel, relb))
    print("%-24s %9.1f %9.1f %9.1f" % ("sum", tot[0], tot[2], tot[1]))


if __name__ == "__main__":
    main()
