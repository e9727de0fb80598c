"""Time uncl_gcn_knn at the bench shape (200 samples x 144 nodes x 256 channels): python tools/knn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncltmo_amd import _hip
lib = _hip.lib()
n = int(os.environ.get("N", "200"))
for dt, code in ((torch.bfloat16, _hip.BF16), (torch.float32, _hip.F32)):
    x = torch.randn(n, 144, 256, device="cuda").to(dt)
    rel = torch.randn(144, 144, device="cuda") * 0.01
    idx = torch.empty(n, 144, 9, dtype=torch.int32, device="cuda")
    for _ in range(3):
        _hip.check(lib.uncl_gcn_knn(x.data_ptr(), code, rel.data_ptr(), idx.data_ptr(), None, n, 144, 256, 9, None, _hip.stream_ptr()), "knn")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.uncl_gcn_knn(x.data_ptr(), code, rel.data_ptr(), idx.data_ptr(), None, n, 144, 256, 9, None, _hip.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    xn = torch.nn.functional.normalize(x.float(), dim=-1)
    d = (xn * xn).sum(-1, keepdim=True) - 2 * xn @ xn.transpose(1, 2) + (xn * xn).sum(-1).unsqueeze(1) + rel
    ref = d.topk(9, dim=-1, largest=False).indices
    print(str(dt), "%.1f us per launch, index agreement with torch %.5f" % (e0.elapsed_time(e1) / 10 * 1e3, (ref == idx.long()).float().mean().item()))
