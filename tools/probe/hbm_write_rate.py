#!/usr/bin/env python
"""HBM write / copy rate of the box, for pricing write-heavy kernels (k_edge_hidden_mm writes 215 MB and reads 125 MB per large launch):
fill and copy of buffers of that size through torch (HIP runtime kernels), best of 20 after warm-up."""
import torch

dev = torch.device("cuda:0")
for mb in (64, 215, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    res = {}
    for name, fn, nbytes in (("fill", lambda: a.fill_(1.0), 4 * n), ("copy", lambda: b.copy_(a), 8 * n), ("axpy", lambda: b.add_(a), 12 * n)):
        for _ in range(5):
            fn()
        best = 1e9
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[name] = nbytes / best / 1e9
    print(f"{mb:5d} MB buffers: " + "  ".join(f"{k} {v:6.2f} TB/s" for k, v in res.items()))
