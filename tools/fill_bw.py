#!/usr/bin/env python3
"""How fast can 805 MB (an 8192^2 f32 RGB frame) be zero-filled?  (development: the floor of a mostly-sky frame)"""
import torch
n = 8192 * 8192 * 3
x = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in (("torch zero_", lambda: x.zero_()), ("torch fill_(1)", lambda: x.fill_(1.0))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-16s %.3f ms  %.2f TB/s" % (name, ms, n * 4 / ms / 1e9))
