import sys, ctypes as C, torch
from pathlib import Path
sys.path.insert(0, '.')
from adaface_amd import _lib, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
for (B, N) in ((6, 6144),):
    x = rn(B, N, 320); gamma = rn(320) * 0.2 + 1; beta = rn(320) * 0.2; wq = rn(320, 320) * 320 ** -0.5; wo = rn(320, 320) * 320 ** -0.5
    bo = rn(320) * 0.1; kv = rn(B, 77, 640)
    y0, p0 = ops.xattn_fused(x, gamma, beta, wq, kv, wo, bo)
    for i in range(400):
        junk = torch.randn(1 + (i * 7919) % 400000, device=dev).sum()
        y1, p1 = ops.xattn_fused(x, gamma, beta, wq, kv, wo, bo)
        if not torch.equal(y0, y1):
            d = (y0 - y1).abs().reshape(B * N, 320)
            rows = (d.sum(dim=1) > 0).nonzero().flatten().tolist()
            print("iter", i, "rows", [(r // 256, r % 256) for r in rows][:20], "n", len(rows), flush=True)
            for r in rows[:3]:
                cols = (d[r] > 0).nonzero().flatten().tolist()
                print("   row", r, "ncols", len(cols), "cols", cols[:40], flush=True)
