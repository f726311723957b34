"""Lab: what a kernel boundary costs.  The same 3x3 convolution launched back to back, timed (a) by one event pair around
200 launches, (b) by an event pair around every launch (what bench_shapes.py / the in-library profiler do)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from adaface_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for cin, cout, hw in ((320, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8)):
    x = torch.nn.functional.silu(torch.randn(16, cin, hw, hw, device=dev, generator=g))
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * (cin * 9) ** -0.5
    b = torch.randn(cout, device=dev, generator=g)
    for _ in range(30):
        ops.conv2d(x, w, b, dtype="bf16")
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.conv2d(x, w, b, dtype="bf16")
        e1.record()
        torch.cuda.synchronize()
        a = e0.elapsed_time(e1) / 200 * 1e3
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for s, e in evs:
            s.record()
            ops.conv2d(x, w, b, dtype="bf16")
            e.record()
        torch.cuda.synchronize()
        bb = sum(s.elapsed_time(e) for s, e in evs) / 200 * 1e3
        res.append((a, bb))
    print(f"conv3x3 {cin}->{cout}@{hw}: " + "  ".join(f"stream {a:.1f} us / bracketed {bb:.1f} us" for a, bb in res), flush=True)
