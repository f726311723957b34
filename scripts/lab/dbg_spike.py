import sys, math, torch
sys.path.insert(0, ".")
from adaface_amd import ops, _lib
dev = torch.device("cuda:0")
def q_(t): return t.to(torch.bfloat16).float()
for ring in (1, 0):
    _lib.set_knob("attn_ring", ring)
    for (qi, kj) in [(100, 10), (100, 70), (100, 330), (100, 342), (100, 470), (100, 458), (17, 470), (300, 330), (300, 342)]:
        g = torch.Generator().manual_seed(6)
        B, N, heads, dh = 1, 512, 8, 40
        q = q_(torch.randn(B, N, heads * dh, generator=g)); k = q_(torch.randn(B, N, heads * dh, generator=g)); v = q_(torch.randn(B, N, heads * dh, generator=g))
        k[:, kj] = q_(q[:, qi] * 24.0)
        got = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, dtype="bf16").cpu()
        bad = ~torch.isfinite(got)
        qs = bad.any(dim=-1)[0].nonzero().flatten().tolist()
        sim = torch.einsum("bihd,bjhd->bhij", q.view(B, N, heads, dh), k.view(B, N, heads, dh)) * dh ** -0.5 * math.log2(math.e)
        print(f"ring {ring} spike q{qi} k{kj} (tile {kj//64}, row {kj%32}, half {(kj%32>>2)&1}): nonfinite {int(bad.sum())} queries {qs[:10]}; score {sim[0,:,qi,kj].tolist()[:3]}")
