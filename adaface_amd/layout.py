"""Host-side parameter inventory of the two networks on the path.

Derives the state_dict key -> shape table from the constructor arguments, following
UNetModel.__init__ (ldm/modules/diffusionmodules/openaimodel.py:517-697) and
Decoder.__init__ / AutoencoderKL.__init__ (ldm/modules/diffusionmodules/model.py:502-573,
ldm/models/autoencoder.py:286-309), so that the Python classes can expose the reference's
parameter names (the weights contract of SURVEY.md §8b) without a GPU being present.
The C library derives the same table independently (af_tensor_name / af_tensor_shape);
tests/test_model_gpu.py checks the two agree.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

Shape = Tuple[int, ...]


def unet_blocks(model_channels: int, channel_mult: Sequence[int], num_res_blocks: int,
                attention_resolutions: Sequence[int], in_channels: int):
    """(input_blocks, middle_block, output_blocks) as lists of layer descriptors
    ("conv_in", cin, cout) | ("res", cin, cout) | ("xfmr", ch) | ("down", ch) | ("up", ch)."""
    mc = model_channels
    inputs: List[list] = [[("conv_in", in_channels, mc)]]
    skip_chans = [mc]
    ch, ds = mc, 1
    nlev = len(channel_mult)
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in attention_resolutions:
                layers.append(("xfmr", ch))
            inputs.append(layers)
            skip_chans.append(ch)
        if level != nlev - 1:
            inputs.append([("down", ch)])
            skip_chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("xfmr", ch), ("res", ch, ch)]
    outputs: List[list] = []
    for level in reversed(range(nlev)):
        mult = channel_mult[level]
        for i in range(num_res_blocks + 1):
            layers = [("res", ch + skip_chans.pop(), mc * mult)]
            ch = mc * mult
            if ds in attention_resolutions:
                layers.append(("xfmr", ch))
            if level and i == num_res_blocks:
                layers.append(("up", ch))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def unet_param_shapes(*, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                      channel_mult, context_dim, transformer_depth=1, **_unused) -> Dict[str, Shape]:
    mc, ted = model_channels, 4 * model_channels
    shapes: Dict[str, Shape] = {}

    def wb(name, w_shape):
        shapes[name + ".weight"] = tuple(w_shape)
        shapes[name + ".bias"] = (w_shape[0],)

    def add_layer(p, desc):
        kind = desc[0]
        if kind == "conv_in":
            wb(p, (desc[2], desc[1], 3, 3))
        elif kind == "res":
            cin, cout = desc[1], desc[2]
            wb(p + ".in_layers.0", (cin,))
            wb(p + ".in_layers.2", (cout, cin, 3, 3))
            wb(p + ".emb_layers.1", (cout, ted))
            wb(p + ".out_layers.0", (cout,))
            wb(p + ".out_layers.3", (cout, cout, 3, 3))
            if cin != cout:
                wb(p + ".skip_connection", (cout, cin, 1, 1))
        elif kind == "xfmr":
            c = desc[1]
            wb(p + ".norm", (c,))
            wb(p + ".proj_in", (c, c, 1, 1))
            for d in range(transformer_depth):
                t = f"{p}.transformer_blocks.{d}"
                for attn, kdim in (("attn1", c), ("attn2", context_dim)):
                    shapes[f"{t}.{attn}.to_q.weight"] = (c, c)
                    shapes[f"{t}.{attn}.to_k.weight"] = (c, kdim)
                    shapes[f"{t}.{attn}.to_v.weight"] = (c, kdim)
                    wb(f"{t}.{attn}.to_out.0", (c, c))
                wb(f"{t}.ff.net.0.proj", (8 * c, c))
                wb(f"{t}.ff.net.2", (c, 4 * c))
                for n in ("norm1", "norm2", "norm3"):
                    wb(f"{t}.{n}", (c,))
            wb(p + ".proj_out", (c, c, 1, 1))
        elif kind == "down":
            wb(p + ".op", (desc[1], desc[1], 3, 3))
        elif kind == "up":
            wb(p + ".conv", (desc[1], desc[1], 3, 3))

    wb("time_embed.0", (ted, mc))
    wb("time_embed.2", (ted, ted))
    inputs, middle, outputs = unet_blocks(mc, channel_mult, num_res_blocks, attention_resolutions, in_channels)
    for i, layers in enumerate(inputs):
        for j, d in enumerate(layers):
            add_layer(f"input_blocks.{i}.{j}", d)
    for j, d in enumerate(middle):
        add_layer(f"middle_block.{j}", d)
    for i, layers in enumerate(outputs):
        for j, d in enumerate(layers):
            add_layer(f"output_blocks.{i}.{j}", d)
    wb("out.0", (mc,))
    wb("out.2", (out_channels, mc, 3, 3))
    return shapes


# tensors the reference zero-initialises (zero_module: openaimodel.py:233,696; attention.py:313)
def unet_zero_init_names(shapes: Dict[str, Shape]) -> List[str]:
    out = []
    for k in shapes:
        stem = k.rsplit(".", 1)[0]
        if stem.endswith(".out_layers.3") or stem.endswith(".proj_out") or stem == "out.2":
            out.append(k)
    return out


def vae_decoder_param_shapes(*, ch, out_ch, ch_mult, num_res_blocks, z_channels, **_unused) -> Dict[str, Shape]:
    """Keys relative to `first_stage_model.decoder.`."""
    shapes: Dict[str, Shape] = {}

    def wb(name, w_shape):
        shapes[name + ".weight"] = tuple(w_shape)
        shapes[name + ".bias"] = (w_shape[0],)

    def res(p, cin, cout):
        wb(p + ".norm1", (cin,))
        wb(p + ".conv1", (cout, cin, 3, 3))
        wb(p + ".norm2", (cout,))
        wb(p + ".conv2", (cout, cout, 3, 3))
        if cin != cout:
            wb(p + ".nin_shortcut", (cout, cin, 1, 1))

    nres = len(ch_mult)
    block_in = ch * ch_mult[-1]
    wb("conv_in", (block_in, z_channels, 3, 3))
    res("mid.block_1", block_in, block_in)
    wb("mid.attn_1.norm", (block_in,))
    for n in ("q", "k", "v", "proj_out"):
        wb("mid.attn_1." + n, (block_in, block_in, 1, 1))
    res("mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * ch_mult[lvl]
        for i in range(num_res_blocks + 1):
            res(f"up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            wb(f"up.{lvl}.upsample.conv", (block_in, block_in, 3, 3))
    wb("norm_out", (block_in,))
    wb("conv_out", (out_ch, block_in, 3, 3))
    return shapes


def vae_encoder_param_shapes(*, ch, ch_mult, num_res_blocks, z_channels, in_channels=3, **_unused) -> Dict[str, Shape]:
    """Keys relative to `first_stage_model.encoder.` (Encoder.__init__, model.py:408-470)."""
    shapes: Dict[str, Shape] = {}

    def wb(name, w_shape):
        shapes[name + ".weight"] = tuple(w_shape)
        shapes[name + ".bias"] = (w_shape[0],)

    def res(p, cin, cout):
        wb(p + ".norm1", (cin,))
        wb(p + ".conv1", (cout, cin, 3, 3))
        wb(p + ".norm2", (cout,))
        wb(p + ".conv2", (cout, cout, 3, 3))
        if cin != cout:
            wb(p + ".nin_shortcut", (cout, cin, 1, 1))

    nres = len(ch_mult)
    wb("conv_in", (ch, in_channels, 3, 3))
    block_in = ch
    for lvl in range(nres):
        block_out = ch * ch_mult[lvl]
        for i in range(num_res_blocks):
            res(f"down.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != nres - 1:
            wb(f"down.{lvl}.downsample.conv", (block_in, block_in, 3, 3))
    res("mid.block_1", block_in, block_in)
    wb("mid.attn_1.norm", (block_in,))
    for n in ("q", "k", "v", "proj_out"):
        wb("mid.attn_1." + n, (block_in, block_in, 1, 1))
    res("mid.block_2", block_in, block_in)
    wb("norm_out", (block_in,))
    wb("conv_out", (2 * z_channels, block_in, 3, 3))
    return shapes


def clip_text_param_shapes(vocab: int, hidden: int, layers: int, heads: int, intermediate: int, max_pos: int) -> Dict[str, Shape]:
    """state_dict keys of transformers CLIPTextModel.text_model (the reference's cond_stage_model.transformer.text_model,
    encoders/modules.py:185), relative to `text_model.`."""
    D, Fm = hidden, intermediate
    out: Dict[str, Shape] = {"embeddings.token_embedding.weight": (vocab, D),
                             "embeddings.position_embedding.weight": (max_pos, D)}
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"] = (D, D)
            out[p + f"self_attn.{n}.bias"] = (D,)
        for n in ("layer_norm1", "layer_norm2"):
            out[p + n + ".weight"] = (D,)
            out[p + n + ".bias"] = (D,)
        out[p + "mlp.fc1.weight"] = (Fm, D)
        out[p + "mlp.fc1.bias"] = (Fm,)
        out[p + "mlp.fc2.weight"] = (D, Fm)
        out[p + "mlp.fc2.bias"] = (D,)
    out["final_layer_norm.weight"] = (D,)
    out["final_layer_norm.bias"] = (D,)
    return out
