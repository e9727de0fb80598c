"""Per-step device allocations / reserved memory of the image training step (run on the GPU box): a steady state must show
zero device allocations -- growth means tensors kept alive by a reference cycle until the cyclic GC runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
a = bench.parse(["--mode", "train"]) if hasattr(bench, "parse") else None
rk = bench.Ranks(a)
tr, step, n = bench.make_trainer(rk, False)
prev = torch.cuda.memory_stats()
for i in range(14):
    step(); torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(i, "dev_alloc", st["num_device_alloc"] - prev["num_device_alloc"], "reserved MB", st["reserved_bytes.all.current"] >> 20,
          "alloc MB", st["allocated_bytes.all.current"] >> 20, "active blocks", st["active.all.current"],
          "large-pool reserved", st["reserved_bytes.large_pool.current"] >> 20, "small", st["reserved_bytes.small_pool.current"] >> 20)
    prev = st
