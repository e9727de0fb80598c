"""Time one 1x1 convolution of the graph block (bf16, M = N*144 pixels) through uncl_conv_igemm: python tools/c1_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import _hip
lib = _hip.lib()
n = int(os.environ.get("N", "200"))
for cin, cout in [(256, 256), (512, 256)]:
    x = torch.randn(n, 12, 12, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, cin, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(cout, device="cuda")
    out = torch.empty(n, 12, 12, cout, device="cuda", dtype=torch.bfloat16)
    d = _hip.ConvDesc()
    d.dtype, d.ksize, d.pad, d.N, d.H, d.W, d.Cin, d.Cout = _hip.BF16, 1, 0, n, 12, 12, cin, cout
    d.src0, d.src0_H, d.src0_W, d.src0_C = x.data_ptr(), 12, 12, cin
    d.weight, d.bias, d.act = w.data_ptr(), b.data_ptr(), _hip.ACT_NONE
    d.out, d.out_H, d.out_W, d.out_C = out.data_ptr(), 12, 12, cout
    for _ in range(3):
        _hip.check(lib.uncl_conv_igemm(C.byref(d), _hip.stream_ptr()), "igemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.uncl_conv_igemm(C.byref(d), _hip.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ref = (x.float().reshape(-1, cin) @ w.float().t()).reshape(out.shape)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    print("Cin %d Cout %d: %.1f us per launch, rel err %.2e" % (cin, cout, e0.elapsed_time(e1) / 20 * 1e3, err))
