"""Seeded parameters for the per-op / per-module goldens (shared by gen_golden.py and the tests; plain torch)."""
import torch


def seeded_params(shapes, seed):
    """{name: shape} -> {name: tensor}: in sorted-name order, randn from one generator; matrices / conv kernels scaled by
    fan_in^-0.5, norm scales ~ 1 + 0.1 N, other vectors ~ 0.1 N (zero-initialised reference tensors get values too)."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        t = torch.randn(shp, generator=g)
        if len(shp) > 1:
            t = t * (float(torch.tensor(shp[1:]).prod()) ** -0.5)
        elif name.endswith("weight"):
            t = 1.0 + 0.1 * t
        else:
            t = 0.1 * t
        out[name] = t
    return out


def _res(cin, cout):
    d = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,), "in_layers.2.weight": (cout, cin, 3, 3),
         "in_layers.2.bias": (cout,), "emb_layers.1.weight": (cout, 256), "emb_layers.1.bias": (cout,),
         "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,), "out_layers.3.weight": (cout, cout, 3, 3),
         "out_layers.3.bias": (cout,)}
    if cin != cout:
        d.update({"skip_connection.weight": (cout, cin, 1, 1), "skip_connection.bias": (cout,)})
    return d


def _attn(dim, ctx, inner):
    return {"to_q.weight": (inner, dim), "to_k.weight": (inner, ctx), "to_v.weight": (inner, ctx),
            "to_out.0.weight": (dim, inner), "to_out.0.bias": (dim,)}


def _ff(dim):
    return {"net.0.proj.weight": (8 * dim, dim), "net.0.proj.bias": (8 * dim,), "net.2.weight": (dim, 4 * dim),
            "net.2.bias": (dim,)}


def _st(c, ctx):
    d = {"norm.weight": (c,), "norm.bias": (c,), "proj_in.weight": (c, c, 1, 1), "proj_in.bias": (c,),
         "proj_out.weight": (c, c, 1, 1), "proj_out.bias": (c,)}
    t = "transformer_blocks.0."
    d.update({t + "attn1." + k: v for k, v in _attn(c, c, c).items()})
    d.update({t + "attn2." + k: v for k, v in _attn(c, ctx, c).items()})
    d.update({t + "ff." + k: v for k, v in _ff(c).items()})
    for n in ("norm1", "norm2", "norm3"):
        d.update({t + n + ".weight": (c,), t + n + ".bias": (c,)})
    return d


# (state_dict shapes of the reference module, seed) per golden, as gen_golden.py instantiated them
MODULES = {
    "gn32": ({"weight": (64,), "bias": (64,)}, 1),
    "normalize": ({"weight": (64,), "bias": (64,)}, 2),
    "ln": ({"weight": (128,), "bias": (128,)}, 3),
    "ff": (_ff(128), 4),
    "attn_self": (_attn(1280, 1280, 1280), 5),
    "attn_cross": (_attn(128, 64, 128), 6),
    "res_same": (_res(64, 64), 7),
    "res_widen": (_res(64, 128), 8),
    "down": ({"op.weight": (64, 64, 3, 3), "op.bias": (64,)}, 9),
    "up": ({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, 10),
    "st": (_st(64, 64), 11),
}


def module_params(name, prefix=""):
    shapes, seed = MODULES[name]
    return {prefix + k: v for k, v in seeded_params(shapes, seed).items()}
