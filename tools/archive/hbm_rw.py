#!/usr/bin/env python3
"""tools/hbm_rw.py — what this box's HBM sustains for pure writes, pure reads and copies (torch kernels on 8 GiB; GB/s).
Context for the roofline fractions: the row-writing kernels are priced against 8 TB/s, the spec figure for reads."""
import torch

dev = torch.device("cuda", 0)
n = 2 << 30  # floats: 8 GiB
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = timed(lambda: x.fill_(1.0))
print(f"fill  (write 8 GiB):        {ms:7.3f} ms  {n * 4 / ms / 1e6:8.0f} GB/s")
ms = timed(lambda: x.zero_())
print(f"zero  (memset 8 GiB):       {ms:7.3f} ms  {n * 4 / ms / 1e6:8.0f} GB/s")
ms = timed(lambda: torch.sum(x))
print(f"sum   (read 8 GiB):         {ms:7.3f} ms  {n * 4 / ms / 1e6:8.0f} GB/s")
ms = timed(lambda: y.copy_(x))
print(f"copy  (read + write 8 GiB): {ms:7.3f} ms  {2 * n * 4 / ms / 1e6:8.0f} GB/s (both directions)")
