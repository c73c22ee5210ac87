#!/usr/bin/env python3
"""tools/hbm_write_probe.py -- what a pure streaming write reaches on this GPU (torch fill / copy of a 6.4 GB tensor):
the ceiling the sample write-back variants (chain STORE, stress FULL) are to be read against."""
import json
import torch

dev = torch.device("cuda:0")
x = torch.empty((32, 3, 1 << 24), dtype=torch.float32, device=dev)
y = torch.empty_like(x)
for name, fn, nbytes in (("fill", lambda: x.fill_(1.0), x.numel() * 4), ("copy (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    print(json.dumps({"op": name, "bytes": nbytes, "ms": ms, "GBps": nbytes / (ms * 1e-3) / 1e9}))
