import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ldm_oracle as O
from adaface_amd.engine import Engine
from tests.test_model_gpu import _unet_kwargs
g = dict(np.load('tests/golden/golden_tiny.npz'))
cfg = O.TINY_UNET
sd = O.synth_state_dict(O.unet_param_shapes(cfg), seed=11)
dev = torch.device('cuda:0')
eng = Engine(dtype=sys.argv[1] if len(sys.argv) > 1 else 'bf16', unet=_unet_kwargs(cfg))
eng.load_state_dict(sd)
x = torch.tensor(g['tiny_x'], device=dev); t = torch.tensor(g['tiny_t'], device=dev); ctx = torch.tensor(g['tiny_ctx'], device=dev)
eng.set_context(ctx, 2, True)
outs = [eng.unet_forward(x, t).cpu().numpy() for _ in range(4)]
for i in range(1, 4):
    d = np.abs(outs[i] - outs[0]); print(i, 'maxdiff', d.max(), 'ndiff', (d > 0).sum(), 'of', d.size)
