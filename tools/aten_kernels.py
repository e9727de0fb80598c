"""Which Python lines of the training step call PyTorch's own device ops (fills, copies, adds ...: ~80 small kernels of ~4 us
per image step)?  One eager step under a TorchDispatchMode; aten ops on device tensors grouped by the innermost frame inside
the package.   python tools/aten_kernels.py [video]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNCL_TRAIN_GRAPH"] = "0"
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
import bench  # noqa: E402

SKIP = ("aten::detach", "aten::view", "aten::reshape", "aten::_unsafe_view", "aten::alias", "aten::empty", "aten::as_strided",
        "aten::select", "aten::slice", "aten::t", "aten::transpose", "aten::expand", "aten::unsqueeze", "aten::squeeze",
        "aten::permute", "aten::_local_scalar_dense", "aten::is_", "aten::new_empty", "aten::lift_fresh", "aten::_to_copy_meta")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name() if hasattr(func, "name") else str(func)
        on_dev = any(isinstance(x, torch.Tensor) and x.is_cuda for x in list(args) + list((kwargs or {}).values()))
        if on_dev and not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if ("uncltmo_amd" in f.filename or f.filename.endswith("bench.py"))
                  and "tools/" not in f.filename]
            where = "%s:%d %s" % (os.path.relpath(fr[-1].filename, ROOT), fr[-1].lineno, fr[-1].name) if fr else "(autograd engine)"
            self.rows[(name, where)] += 1
        return func(*args, **(kwargs or {}))


video = len(sys.argv) > 1 and sys.argv[1] == "video"
a = bench.parse([])
rk = bench.Ranks(a)
tr, step, _ = bench.make_trainer(rk, video)
for _ in range(3):
    step()
torch.cuda.synchronize()
log = Log()
with log:
    step()
torch.cuda.synchronize()
print("aten calls on device tensors in one step: %d" % sum(log.rows.values()))
for (name, where), n in sorted(log.rows.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-26s %s" % (n, name, where))
