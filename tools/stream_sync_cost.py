#!/usr/bin/env python
"""Cost of a cross-stream dependency on this platform: a chain of small kernels on one stream vs the same kernels
alternating between two streams with an event wait at every hand-over."""
import torch
x = torch.zeros(1 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(n, two):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(s1)
    for i in range(n):
        if two:
            a, b = (s1, s2) if i % 2 == 0 else (s2, s1)
            with torch.cuda.stream(b):
                b.wait_stream(a)
                x.add_(1.0)
        else:
            with torch.cuda.stream(s1):
                x.add_(1.0)
    s1.wait_stream(s2)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for _ in range(2):
    print(f"one stream: {run(200, False):6.1f} us per kernel;  two streams, a wait per kernel: {run(200, True):6.1f} us per kernel")
