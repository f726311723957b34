import sys, math, torch
sys.path.insert(0, "/root/repo")
from adaface_amd import _lib, ops
_lib.load(); _lib.set_knob("plan_log", 1)
dev = torch.device("cuda:0")
for B in (5, 6, 10, 12, 14):
    x = torch.randn(B, 1280, 8, 8, device=dev); w = torch.randn(1280, 1280, 3, 3, device=dev) * 0.01
    print("B", B, file=sys.stderr, flush=True); ops.conv2d(x, w, None, dtype="bf16")
for M in (1000, 1100, 2000):
    x = torch.randn(M, 5120, device=dev); w = torch.randn(1280, 5120, device=dev) * 0.01
    print("M", M, file=sys.stderr, flush=True); ops.linear(x, w, None, None, dtype="bf16")
