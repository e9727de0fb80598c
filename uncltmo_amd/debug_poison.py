"""Debug aid (never on the product path): fill the caching allocator's FREE memory with a pattern, so that a kernel which reads
a workspace before writing it sees the pattern instead of the zeros a fresh process happens to get from the driver.

0x7F bytes: as bf16 / fp32 3.4e38 (finite, absurd), as int32 2139062143 -- an index read from unwritten memory lands far outside
any tensor and faults instead of passing by luck.  tests/conftest.py runs it before every GPU test when UNCL_POISON_GB is set;
bench.py before its training legs.
"""
import os

import torch


def poison_free_memory(gb=None, small_blocks=1024, byte=0x7F):
    gb = float(os.environ.get("UNCL_POISON_GB", "0")) if gb is None else gb
    if gb <= 0 or not torch.cuda.is_available():
        return 0
    torch.cuda.synchronize()
    held = []
    # what is cached now (whatever its block sizes): take it in decreasing sizes until the cache is exhausted, then one big
    # block that becomes the arena later large requests are carved from
    big = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda").fill_(byte)
    held.append(big)
    for sz in (1 << 24, 1 << 22, 1 << 20):
        for _ in range(64):
            held.append(torch.empty(sz, dtype=torch.uint8, device="cuda").fill_(byte))
    for _ in range(small_blocks):                 # the small pool (requests under 1 MiB live in 2 MiB blocks of their own)
        held.append(torch.empty(1 << 19, dtype=torch.uint8, device="cuda").fill_(byte))
    n = sum(t.numel() for t in held)
    torch.cuda.synchronize()
    del held, big
    return n
